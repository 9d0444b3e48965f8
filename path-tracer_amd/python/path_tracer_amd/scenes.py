"""Deterministic procedural stand-ins for the reference's scenes.

The reference loads Sponza / Bistro glTF assets (src/scene.cpp:76) that are Git-LFS pointers in its
repository, so none of them is available offline.  These generators emit geometry of the same class
(an enclosed colonnaded courtyard, a street between buildings) already flattened the way
src/scene.cpp:84-176 flattens a glTF: tightly packed positions / normals / uvs, 3 packed index words
per triangle (24-bit vertex index | one byte of the 24-bit material id, src/scene.cpp:140-142), one
48-byte gpu_material per primitive (include/material.hpp:17-25) and RGBA8 textures.

Everything is integer-hash / numpy arithmetic on a fixed seed, so a scene is a pure function of
(kind, target_triangles, seed)."""
import numpy as np

MATERIAL_DTYPE = np.dtype([("base_color", "<f4", 4), ("albedo_tex", "<u4"), ("albedo_sampler", "<u4"),
                           ("mr_tex", "<u4"), ("mr_sampler", "<u4"), ("metalness", "<f4"), ("roughness", "<f4"),
                           ("pad", "<f4", 2)])
assert MATERIAL_DTYPE.itemsize == 48
# the shader's `sphere` struct (compute.comp:31-36): vec4 position, material, float radius, padded to a multiple of 16 bytes
SPHERE_DTYPE = np.dtype([("position", "<f4", 4), ("material", MATERIAL_DTYPE), ("radius", "<f4"), ("pad", "<f4", 3)])
assert SPHERE_DTYPE.itemsize == 80

MAX_PRIMITIVE_VERTICES = 65535  # the reference's loader reads u16 indices only (include/mesh.hpp:87)


def _hash_u32(x):
    x = np.asarray(x, dtype=np.uint64)
    x = (x ^ (x >> 16)) * np.uint64(0x7FEB352D) & np.uint64(0xFFFFFFFF)
    x = (x ^ (x >> 15)) * np.uint64(0x846CA68B) & np.uint64(0xFFFFFFFF)
    x = x ^ (x >> 16)
    return x.astype(np.uint32)


def _noise01(idx, seed):
    return _hash_u32(np.asarray(idx, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(seed)).astype(np.float64) / 4294967296.0


def make_texture(size, seed, kind):
    """RGBA8 texture.  kind 'albedo': coloured value-noise + checker; 'mr': R=metalness map, G=roughness map."""
    y, x = np.mgrid[0:size, 0:size]
    cell = 1 << (2 + seed % 4)
    lattice = _noise01((y // cell) * 4099 + (x // cell), seed)
    fine = _noise01(y * size + x, seed * 7 + 1)
    checker = (((x // (size // 8)) + (y // (size // 8))) & 1).astype(np.float64)
    out = np.zeros((size, size, 4), dtype=np.uint8)
    if kind == "albedo":
        tint = 0.35 + 0.6 * _noise01(np.arange(3), seed * 31 + 5)
        v = 0.55 * lattice + 0.25 * fine + 0.2 * checker
        for ch in range(3):
            out[..., ch] = np.clip((0.25 + 0.75 * v) * tint[ch] * 255.0, 0, 255).astype(np.uint8)
        out[..., 3] = 255
    else:
        out[..., 0] = np.clip((0.15 * lattice + 0.1 * checker) * 255.0, 0, 255).astype(np.uint8)
        out[..., 1] = np.clip((0.35 + 0.5 * lattice + 0.15 * fine) * 255.0, 0, 255).astype(np.uint8)
        out[..., 2] = 0
        out[..., 3] = 255
    return out


class _Builder:
    def __init__(self, seed, dry=False):
        self.seed = seed
        self.dry = dry            # count triangles only
        self.triangle_count = 0
        self.positions, self.normals, self.uvs, self.indices = [], [], [], []
        self.materials = []
        self.vertex_count = 0
        self.prim_count = 0

    def _jitter(self, p, amount):
        n = p.shape[0]
        base = (self.vertex_count + np.arange(n, dtype=np.uint64)) * np.uint64(3)
        j = np.stack([_noise01(base + np.uint64(k), self.seed + 17) for k in range(3)], axis=1) - 0.5
        return p + j * amount

    def add_patch(self, fn, nu, nv, material, uv_scale=(1.0, 1.0), jitter=1e-3, flip=False):
        """Tessellated parametric patch: fn(u, v) -> (N,3) positions for u, v in [0,1]; split to fit u16 indices."""
        self.triangle_count += 2 * nu * nv
        if self.dry:
            return
        rows_per = max(1, min(nv, MAX_PRIMITIVE_VERTICES // (nu + 1) - 1))
        v0 = 0
        while v0 < nv:
            v1 = min(nv, v0 + rows_per)
            self._add_patch_rows(fn, nu, nv, v0, v1, material, uv_scale, jitter, flip)
            v0 = v1

    def _add_patch_rows(self, fn, nu, nv, v0, v1, material, uv_scale, jitter, flip):
        us = np.linspace(0.0, 1.0, nu + 1)
        vs = np.arange(v0, v1 + 1) / float(nv)
        U, V = np.meshgrid(us, vs, indexing="xy")  # shape (rows, nu+1)
        u, v = U.ravel(), V.ravel()
        p = fn(u, v)
        eps = 1e-3
        du = fn(np.clip(u + eps, 0, 1), v) - fn(np.clip(u - eps, 0, 1), v)
        dv = fn(u, np.clip(v + eps, 0, 1)) - fn(u, np.clip(v - eps, 0, 1))
        n = np.cross(du, dv)
        if flip:
            n = -n
        ln = np.linalg.norm(n, axis=1, keepdims=True)
        n = np.where(ln > 1e-20, n / np.maximum(ln, 1e-20), np.array([[0.0, 1.0, 0.0]]))
        # jitter only interior vertices of the full patch so that split pieces stay watertight
        interior = ((u > 0) & (u < 1) & (v > 0) & (v < 1))[:, None]
        p = np.where(interior, self._jitter(p, jitter), p)
        rows = v1 - v0
        r, c = np.mgrid[0:rows, 0:nu]
        a = (r * (nu + 1) + c).ravel()
        b = a + 1
        d = a + (nu + 1)
        e = d + 1
        if flip:
            tri = np.stack([np.stack([a, d, b], 1), np.stack([b, d, e], 1)], 1).reshape(-1, 3)
        else:
            tri = np.stack([np.stack([a, b, d], 1), np.stack([b, e, d], 1)], 1).reshape(-1, 3)
        self._append(p, n, np.stack([u * uv_scale[0], v * uv_scale[1]], 1), tri, material)

    def _append(self, p, n, uv, tri, material):
        assert p.shape[0] <= MAX_PRIMITIVE_VERTICES
        mat_id = len(self.materials)
        self.materials.append(material)
        packed = (tri + self.vertex_count).astype(np.uint32)
        assert self.vertex_count + p.shape[0] < (1 << 24), "24-bit vertex index overflow (src/scene.cpp:140)"
        assert mat_id < (1 << 24)
        packed[:, 0] |= np.uint32((mat_id << 8) & 0xFF000000)
        packed[:, 1] |= np.uint32((mat_id << 16) & 0xFF000000)
        packed[:, 2] |= np.uint32((mat_id << 24) & 0xFF000000)
        self.positions.append(p.astype(np.float32))
        self.normals.append(n.astype(np.float32))
        self.uvs.append(uv.astype(np.float32))
        self.indices.append(packed)
        self.vertex_count += p.shape[0]
        self.prim_count += 1


def _material(rng, n_albedo, n_mr, mirror=False):
    m = np.zeros((), dtype=MATERIAL_DTYPE)
    m["base_color"] = np.concatenate([0.55 + 0.45 * rng.random(3), [1.0]]).astype(np.float32)
    m["albedo_tex"] = 1 + rng.integers(0, n_albedo)
    m["albedo_sampler"] = 1
    if mirror:
        # metalness 1, roughness 0, untextured -> the shader's perfect-mirror branch (compute.comp:162)
        m["mr_tex"] = 0
        m["mr_sampler"] = 0
        m["metalness"] = 1.0
        m["roughness"] = 0.0
    else:
        m["mr_tex"] = 1 + n_albedo + rng.integers(0, n_mr)
        m["mr_sampler"] = 1
        m["metalness"] = np.float32(rng.choice([0.0, 0.2, 1.0], p=[0.5, 0.3, 0.2]))
        m["roughness"] = np.float32(0.25 + 0.75 * rng.random())
    return m


def _plane(origin, eu, ev, bump=None):
    origin, eu, ev = (np.asarray(a, dtype=np.float64) for a in (origin, eu, ev))
    nrm = np.cross(eu, ev)
    nrm = nrm / np.linalg.norm(nrm)

    def fn(u, v):
        p = origin[None, :] + u[:, None] * eu[None, :] + v[:, None] * ev[None, :]
        if bump is not None:
            p = p + nrm[None, :] * bump(u, v)[:, None]
        return p
    return fn


def _cylinder(center, radius, height, flute=0.0):
    cx, cy, cz = center

    def fn(u, v):
        ang = 2 * np.pi * u
        r = radius * (1.0 + flute * np.cos(16 * ang)) * (1.0 - 0.08 * v)
        return np.stack([cx + r * np.cos(ang), cy + height * v, cz - r * np.sin(ang)], 1)
    return fn


def _sphere(center, radius):
    cx, cy, cz = center

    def fn(u, v):
        th = np.pi * (0.02 + 0.96 * v)
        ph = 2 * np.pi * u
        return np.stack([cx + radius * np.sin(th) * np.cos(ph), cy + radius * np.cos(th), cz + radius * np.sin(th) * np.sin(ph)], 1)
    return fn


def _arch(p0, p1, rise, depth):
    p0 = np.asarray(p0, dtype=np.float64); p1 = np.asarray(p1, dtype=np.float64)
    along = p1 - p0
    side = np.cross(along, [0, 1, 0]); side = side / np.linalg.norm(side) * depth

    def fn(u, v):
        base = p0[None, :] + u[:, None] * along[None, :] + (v[:, None] - 0.5) * side[None, :]
        base[:, 1] += rise * np.sin(np.pi * u)
        return base
    return fn


def _cloth(p0, p1, sag, width, waves):
    p0 = np.asarray(p0, dtype=np.float64); p1 = np.asarray(p1, dtype=np.float64)
    along = p1 - p0
    side = np.cross(along, [0, 1, 0]); side = side / np.linalg.norm(side) * width

    def fn(u, v):
        base = p0[None, :] + u[:, None] * along[None, :] + (v[:, None] - 0.5) * side[None, :]
        base[:, 1] += -sag * np.sin(np.pi * u) + 0.08 * np.sin(waves * 2 * np.pi * u + 5 * v)
        return base
    return fn


def generate(kind="atrium", target_triangles=262144, seed=0xC0FFEE, texture_size=256):
    """Returns a dict of flat scene arrays.  kind: 'atrium' (enclosed courtyard, Sponza-class),
    'street_interior' (mostly enclosed, Bistro-interior-class) or 'street_exterior' (open to the sky)."""
    rng = np.random.default_rng(seed)
    n_albedo, n_mr = 10, 6
    textures = [None]
    for i in range(n_albedo):
        textures.append(make_texture(texture_size, seed + i, "albedo"))
    for i in range(n_mr):
        textures.append(make_texture(texture_size, seed + 100 + i, "mr"))

    # Tessellation scale: a patch of (nu x nv) quads is 2*nu*nv triangles; two dry (count-only) passes
    # solve for the scale `s` that lands on the requested triangle count.
    s = np.sqrt(max(target_triangles, 2048) / 262144.0)
    for _ in range(3):
        dry = _Builder(seed, dry=True)
        _emit(dry, kind, s, np.random.default_rng(seed), n_albedo, n_mr)
        s *= np.sqrt(max(target_triangles, 2048) / max(dry.triangle_count, 1))
    b = _Builder(seed)
    L = _emit(b, kind, s, rng, n_albedo, n_mr)

    positions = np.concatenate(b.positions).astype(np.float32)
    normals = np.concatenate(b.normals).astype(np.float32)
    uvs = np.concatenate(b.uvs).astype(np.float32)
    indices = np.concatenate(b.indices).astype(np.uint32)
    materials = np.array(b.materials, dtype=MATERIAL_DTYPE)

    vid = indices & np.uint32(0x00FFFFFF)
    tri_pos = positions[vid.reshape(-1)].reshape(-1, 9)

    if kind == "atrium":
        cam = dict(pos=[-L * 0.42, 2.2, 0.6], target=[L * 0.3, 3.5, -0.4], v_fov=80.0, aperture=0.1, focus=10.0)
    else:
        cam = dict(pos=[-L * 0.45, 2.0, 1.0], target=[L * 0.35, 4.5, -0.8], v_fov=75.0, aperture=0.1, focus=10.0)
    return dict(kind=kind, positions=positions.reshape(-1), normals=normals.reshape(-1), uvs=uvs.reshape(-1),
                indices=indices.reshape(-1), materials=materials, textures=textures, triangles=np.ascontiguousarray(tri_pos),
                camera=cam, primitive_count=b.prim_count)


def _emit(b, kind, s, rng, n_albedo, n_mr):
    """Lays the scene out into builder `b` at tessellation scale `s`; returns the scene length."""

    def q(n):
        return max(2, int(round(n * s)))

    enclosed = kind in ("atrium", "street_interior")
    L, Wd, Ht = (40.0, 16.0, 12.0) if kind == "atrium" else (90.0, 22.0, 16.0)

    bump_floor = lambda u, v: 0.03 * np.sin(40 * u) * np.sin(40 * v)
    b.add_patch(_plane([-L / 2, 0, -Wd / 2], [L, 0, 0], [0, 0, Wd], bump_floor), q(224), q(96), _material(rng, n_albedo, n_mr),
                uv_scale=(12, 5), flip=True)
    wall_bump = lambda u, v: 0.05 * np.sin(60 * u) * np.cos(24 * v)
    # long walls (facades)
    b.add_patch(_plane([-L / 2, 0, -Wd / 2], [L, 0, 0], [0, Ht, 0], wall_bump), q(224), q(64), _material(rng, n_albedo, n_mr), uv_scale=(10, 3))
    b.add_patch(_plane([-L / 2, 0, Wd / 2], [L, 0, 0], [0, Ht, 0], wall_bump), q(224), q(64), _material(rng, n_albedo, n_mr), uv_scale=(10, 3), flip=True)
    # end walls
    b.add_patch(_plane([-L / 2, 0, -Wd / 2], [0, 0, Wd], [0, Ht, 0], wall_bump), q(96), q(64), _material(rng, n_albedo, n_mr), uv_scale=(4, 3), flip=True)
    b.add_patch(_plane([L / 2, 0, -Wd / 2], [0, 0, Wd], [0, Ht, 0], wall_bump), q(96), q(64), _material(rng, n_albedo, n_mr), uv_scale=(4, 3))
    if enclosed:
        hole = 0.0 if kind == "street_interior" else 0.35  # the atrium keeps a skylight strip
        b.add_patch(_plane([-L / 2, Ht, -Wd / 2], [L, 0, 0], [0, 0, Wd * (0.5 - hole / 2)], wall_bump), q(160), q(32),
                    _material(rng, n_albedo, n_mr), uv_scale=(10, 2))
        b.add_patch(_plane([-L / 2, Ht, Wd * hole / 2], [L, 0, 0], [0, 0, Wd * (0.5 - hole / 2)], wall_bump), q(160), q(32),
                    _material(rng, n_albedo, n_mr), uv_scale=(10, 2))

    # colonnades: two rows of fluted columns, arches between neighbours, an upper gallery floor
    n_cols = 10 if kind == "atrium" else 18
    for side in (-1, 1):
        z = side * Wd * 0.3
        xs = np.linspace(-L / 2 + 2.5, L / 2 - 2.5, n_cols)
        for i, x in enumerate(xs):
            mirror = (i % 5 == 2)
            b.add_patch(_cylinder((x, 0.0, z), 0.45, Ht * 0.45, flute=0.04), q(40), q(36), _material(rng, n_albedo, n_mr, mirror=mirror),
                        uv_scale=(2, 4))
            if i + 1 < len(xs):
                b.add_patch(_arch((x, Ht * 0.45, z), (xs[i + 1], Ht * 0.45, z), 1.1, 0.9), q(36), q(10), _material(rng, n_albedo, n_mr),
                            uv_scale=(2, 1), flip=(side > 0))
        b.add_patch(_plane([-L / 2, Ht * 0.55, z - 1.2 * side], [L, 0, 0], [0, 0, 2.4 * side], bump_floor), q(160), q(12),
                    _material(rng, n_albedo, n_mr), uv_scale=(10, 1), flip=(side < 0))

    # draped cloth across the nave and props on the ground
    n_cloth = 6 if kind == "atrium" else 12
    for i in range(n_cloth):
        x = -L / 2 + (i + 0.5) * L / n_cloth
        b.add_patch(_cloth((x, Ht * 0.52, -Wd * 0.3), (x + 0.8, Ht * 0.52, Wd * 0.3), 1.4, 1.6, 3 + i % 3), q(72), q(20),
                    _material(rng, n_albedo, n_mr), uv_scale=(3, 1))
    n_props = 10 if kind == "atrium" else 28
    for i in range(n_props):
        px = (rng.random() - 0.5) * (L - 6)
        pz = (rng.random() - 0.5) * Wd * 0.45
        r = 0.35 + 0.5 * rng.random()
        b.add_patch(_sphere((px, r, pz), r), q(36), q(24), _material(rng, n_albedo, n_mr, mirror=(i % 4 == 0)), uv_scale=(2, 1), flip=True)
    if kind == "street_exterior":
        # free-standing buildings beyond the street walls + street furniture; open sky above
        for i in range(8):
            x0 = -L / 2 + i * L / 8 + 1.0
            hgt = Ht * (0.6 + 0.8 * rng.random())
            zc = (-1 if i % 2 else 1) * (Wd / 2 + 6.0)
            for (o, eu, ev, fl) in (([x0, 0, zc - 3], [L / 8 - 2, 0, 0], [0, hgt, 0], False),
                                    ([x0, 0, zc + 3], [L / 8 - 2, 0, 0], [0, hgt, 0], True),
                                    ([x0, hgt, zc - 3], [L / 8 - 2, 0, 0], [0, 0, 6], False)):
                b.add_patch(_plane(o, eu, ev, wall_bump), q(48), q(48), _material(rng, n_albedo, n_mr), uv_scale=(3, 3), flip=fl)

    return L


def tiny_scene(seed=1):
    """A few dozen triangles exercising every shading branch: textured diffuse floor, a mirror quad,
    a rough-metal quad and an untextured (texture id 0) triangle.  Small enough for pure-CPU tests."""
    rng = np.random.default_rng(seed)
    textures = [None, make_texture(16, seed, "albedo"), make_texture(8, seed + 1, "mr")]
    b = _Builder(seed)

    def mat(base, alb, mr, metal, rough):
        m = np.zeros((), dtype=MATERIAL_DTYPE)
        m["base_color"] = np.array(list(base) + [1.0], np.float32)
        m["albedo_tex"] = alb; m["albedo_sampler"] = 1 if alb else 0
        m["mr_tex"] = mr; m["mr_sampler"] = 1 if mr else 0
        m["metalness"] = metal; m["roughness"] = rough
        return m
    b.add_patch(_plane([-4, 0, -4], [8, 0, 0], [0, 0, 8]), 3, 3, mat((0.9, 0.8, 0.7), 1, 2, 0.3, 0.8), uv_scale=(2, 2), flip=True)
    b.add_patch(_plane([-4, 0, -4], [8, 0, 0], [0, 5, 0]), 2, 2, mat((1, 1, 1), 1, 0, 1.0, 0.0), uv_scale=(1, 1))  # mirror wall
    b.add_patch(_plane([-4, 0, -4], [0, 0, 8], [0, 5, 0]), 2, 2, mat((0.7, 0.9, 0.6), 1, 2, 1.0, 0.6), flip=True)   # rough metal wall
    b.add_patch(_sphere((0.5, 1.0, 0.2), 1.0), 6, 4, mat((0.8, 0.3, 0.3), 0, 0, 0.0, 1.0), flip=True)               # untextured
    positions = np.concatenate(b.positions).astype(np.float32)
    indices = np.concatenate(b.indices).astype(np.uint32)
    vid = indices & np.uint32(0x00FFFFFF)
    tri_pos = positions[vid.reshape(-1)].reshape(-1, 9)
    _ = rng
    return dict(kind="tiny", positions=positions.reshape(-1), normals=np.concatenate(b.normals).astype(np.float32).reshape(-1),
                uvs=np.concatenate(b.uvs).astype(np.float32).reshape(-1), indices=indices.reshape(-1),
                materials=np.array(b.materials, dtype=MATERIAL_DTYPE), textures=textures,
                triangles=np.ascontiguousarray(tri_pos),
                camera=dict(pos=[3.0, 2.5, 3.5], target=[-1.0, 1.0, -1.0], v_fov=70.0, aperture=0.1, focus=5.0),
                primitive_count=b.prim_count)


def widen_indices(scene):
    """The same scene with the 4-word index layout (v0, v1, v2, material per triangle) instead of the reference's packed
    3 words: the extension for scenes past 2^24 vertices."""
    idx = np.asarray(scene["indices"], np.uint32).reshape(-1, 3)
    mat = ((idx[:, 0] >> 24) << 16) | ((idx[:, 1] >> 24) << 8) | (idx[:, 2] >> 24)
    wide = np.concatenate([idx & np.uint32(0x00FFFFFF), mat[:, None]], axis=1).astype(np.uint32)
    return dict(scene, indices=wide.reshape(-1), index_words=4)


def instance_grid(scene, nx, nz, spacing=None):
    """nx * nz translated copies of `scene` on a grid in x and z, as ONE flat scene in the 4-word index layout (the
    copies share the materials and textures).  This is how a 10^8-triangle scene is made from a 3 * 10^6-triangle one."""
    base = widen_indices(scene) if int(scene.get("index_words", 3)) == 3 else scene
    P = np.asarray(base["positions"], np.float32).reshape(-1, 3)
    N = np.asarray(base["normals"], np.float32).reshape(-1, 3)
    U = np.asarray(base["uvs"], np.float32).reshape(-1, 2)
    I = np.asarray(base["indices"], np.uint32).reshape(-1, 4)
    extent = P.max(axis=0) - P.min(axis=0)
    if spacing is None:
        spacing = (float(extent[0]) * 1.02, float(extent[2]) * 1.02)
    copies = nx * nz
    nv = P.shape[0]
    assert copies * nv < 2 ** 32
    positions = np.empty((copies, nv, 3), np.float32)
    indices = np.empty((copies, I.shape[0], 4), np.uint32)
    k = 0
    for ix in range(nx):
        for iz in range(nz):
            offset = np.array([(ix - (nx - 1) / 2.0) * spacing[0], 0.0, (iz - (nz - 1) / 2.0) * spacing[1]], np.float32)
            positions[k] = P + offset
            indices[k, :, :3] = I[:, :3] + np.uint32(k * nv)
            indices[k, :, 3] = I[:, 3]
            k += 1
    positions = positions.reshape(-1, 3)
    out = dict(base, positions=positions.reshape(-1), normals=np.tile(N, (copies, 1)).reshape(-1), uvs=np.tile(U, (copies, 1)).reshape(-1),
               indices=indices.reshape(-1), index_words=4, kind=f"{scene.get('kind', 'scene')} x{copies}")
    out["triangles"] = np.ascontiguousarray(positions[indices.reshape(-1, 4)[:, :3].reshape(-1)].reshape(-1, 9))
    return out


def concat_scenes(parts):
    """Several scenes as ONE set of arrays in the 4-word index layout, each part's triangles in a run of their own.
    Returns (scene, object_first): object_first holds len(parts) + 1 triangle offsets."""
    positions, normals, uvs, indices, materials, textures, triangles = [], [], [], [], [], [None], []
    object_first = [0]
    nv = nm = 0
    for part in parts:
        w = widen_indices(part) if int(part.get("index_words", 3)) == 3 else part
        I = np.asarray(w["indices"], np.uint32).reshape(-1, 4).copy()
        I[:, :3] += np.uint32(nv)
        I[:, 3] += np.uint32(nm)
        m = np.array(w["materials"], dtype=MATERIAL_DTYPE, copy=True)
        shift = len(textures) - 1                                  # texture numbers are 1-based, 0 = none
        for field in ("albedo_tex", "mr_tex"):
            m[field] = np.where(m[field] > 0, m[field] + shift, 0)
        textures += [t for t in w["textures"][1:]]
        positions.append(np.asarray(w["positions"], np.float32)); normals.append(np.asarray(w["normals"], np.float32))
        uvs.append(np.asarray(w["uvs"], np.float32)); indices.append(I.reshape(-1)); materials.append(m)
        triangles.append(np.asarray(w["triangles"], np.float32).reshape(-1, 9))
        nv += len(w["positions"]) // 3
        nm += len(m)
        object_first.append(object_first[-1] + I.shape[0])
    out = dict(parts[0], kind="+".join(str(p.get("kind", "scene")) for p in parts), positions=np.concatenate(positions), normals=np.concatenate(normals),
               uvs=np.concatenate(uvs), indices=np.concatenate(indices), index_words=4, materials=np.concatenate(materials), textures=textures,
               triangles=np.concatenate(triangles))
    return out, np.array(object_first, np.uint32)


def grid_transforms(scene, nx, nz, spacing=None):
    """The placements of instance_grid() as (nx * nz, 3, 4) object-to-world matrices, in the same order."""
    P = np.asarray(scene["positions"], np.float32).reshape(-1, 3)
    extent = P.max(axis=0) - P.min(axis=0)
    if spacing is None:
        spacing = (float(extent[0]) * 1.02, float(extent[2]) * 1.02)
    m = np.zeros((nx * nz, 3, 4), np.float32)
    m[:, 0, 0] = m[:, 1, 1] = m[:, 2, 2] = 1.0
    k = 0
    for ix in range(nx):
        for iz in range(nz):
            m[k, :, 3] = np.array([(ix - (nx - 1) / 2.0) * spacing[0], 0.0, (iz - (nz - 1) / 2.0) * spacing[1]], np.float32)
            k += 1
    return m


def expand_instances(scene, object_to_world):
    """The placed copies of `scene` written out as ONE flat scene in the 4-word index layout: positions through the
    matrices, normals through their inverse transposes.  Triangle k of copy i becomes triangle i * T + k."""
    base = widen_indices(scene) if int(scene.get("index_words", 3)) == 3 else scene
    M = np.asarray(object_to_world, np.float64).reshape(-1, 3, 4)
    P = np.asarray(base["positions"], np.float64).reshape(-1, 3)
    N = np.asarray(base["normals"], np.float64).reshape(-1, 3)
    U = np.asarray(base["uvs"], np.float32).reshape(-1, 2)
    I = np.asarray(base["indices"], np.uint32).reshape(-1, 4)
    copies, nv = M.shape[0], P.shape[0]
    assert copies * nv < 2 ** 32
    positions = np.empty((copies, nv, 3), np.float32)
    normals = np.empty((copies, nv, 3), np.float32)
    indices = np.empty((copies, I.shape[0], 4), np.uint32)
    for k in range(copies):
        A = M[k, :, :3]
        positions[k] = (P @ A.T + M[k, :, 3]).astype(np.float32)
        n = N @ np.linalg.inv(A)            # rows: (A^-T n)^T
        normals[k] = (n / np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-30)).astype(np.float32)
        indices[k, :, :3] = I[:, :3] + np.uint32(k * nv)
        indices[k, :, 3] = I[:, 3]
    positions = positions.reshape(-1, 3)
    out = dict(base, positions=positions.reshape(-1), normals=normals.reshape(-1), uvs=np.tile(U, (copies, 1)).reshape(-1),
               indices=indices.reshape(-1), index_words=4, kind=f"{scene.get('kind', 'scene')} x{copies}")
    out["triangles"] = np.ascontiguousarray(positions[indices.reshape(-1, 4)[:, :3].reshape(-1)].reshape(-1, 9))
    return out


def pack_metadata(camera_block, width, height, max_bounce=3, min_bounce=1, sample_index=1, enable_dof=0, debug_bvh=0,
                  downscale_factor=1):
    """160-byte scene metadata: the 128-byte camera followed by eight 32-bit words
    (include/scene.hpp:11-27 in the reference; compute.comp:67-79 on the device side)."""
    meta = np.zeros(160, dtype=np.uint8)
    meta[:128] = np.frombuffer(np.ascontiguousarray(camera_block).tobytes(), dtype=np.uint8)[:128]
    u = meta.view(np.uint32)
    u[32:39] = [max_bounce, min_bounce, width, height, sample_index, enable_dof, debug_bvh]
    meta.view(np.int32)[39] = downscale_factor
    return meta


def save_ptrscene(scene, path):
    """Writes the flat arrays in the .ptrscene layout the C++ host reads (host/scene.cpp: scene_source::load)."""
    textures = [t for t in scene["textures"] if t is not None]
    with open(path, "wb") as f:
        n_vertices = len(scene["positions"]) // 3
        f.write(b"PTRS")
        f.write(np.array([2 if int(scene.get("index_words", 3)) == 4 else 1, len(scene["indices"]), n_vertices, len(scene["materials"]), len(textures)], dtype="<u4").tobytes())
        f.write(np.ascontiguousarray(scene["indices"], dtype="<u4").tobytes())
        f.write(np.ascontiguousarray(scene["positions"], dtype="<f4").tobytes())
        f.write(np.ascontiguousarray(scene["normals"], dtype="<f4").tobytes())
        f.write(np.ascontiguousarray(scene["uvs"], dtype="<f4").tobytes())
        f.write(np.ascontiguousarray(scene["materials"]).tobytes())
        for t in textures:
            f.write(np.array([t.shape[1], t.shape[0]], dtype="<u4").tobytes())
            f.write(np.ascontiguousarray(t, dtype=np.uint8).tobytes())
