"""Shared helpers of the test-suite: scene bundles for the oracle and for the GPU path."""
import numpy as np

from path_tracer_amd import host, scenes


def make_bundle(kind="tiny", target=4000, width=64, height=40, seed=None, **meta_kw):
    """Scene arrays + threaded nodes + flattened layout + metadata, ready for OracleScene / GpuScene."""
    if kind == "tiny":
        sc = scenes.tiny_scene(seed or 1)
    else:
        sc = scenes.generate(kind, target, seed=seed or 0xC0FFEE, texture_size=64)
    nodes = host.bvh_build(sc["triangles"])
    ranks = host.bvh_visit_ranks(nodes, sc["triangles"].shape[0])
    flat = host.bvh_flatten(sc["triangles"], ranks)
    c = sc["camera"]
    cam = host.camera(c["pos"], c["target"], c["v_fov"], width / height, c["aperture"], c["focus"])
    meta = scenes.pack_metadata(cam, width, height, **meta_kw)
    return dict(sc, nodes=nodes, ranks=ranks, flat=flat, meta=meta, width=width, height=height)


def add_spheres(bundle, spheres):
    """The bundle with sphere leaves: `spheres` is a list of (centre, radius, material record); the flattened layout is rebuilt
    over the triangles and the spheres, the 80-byte records go to bundle["sphere_records"]."""
    records = np.zeros(len(spheres), dtype=scenes.SPHERE_DTYPE)
    for i, (centre, radius, material) in enumerate(spheres):
        records[i]["position"] = list(centre) + [0.0]
        records[i]["material"] = material
        records[i]["radius"] = radius
    geometry = np.concatenate([records["position"][:, :3], records["radius"][:, None]], axis=1).astype(np.float32)
    flat = host.bvh_flatten(bundle["triangles"], bundle["ranks"], spheres=geometry)
    return dict(bundle, flat=flat, sphere_records=records)


def sphere_tuples(rows9):
    """(n, 9) rows of x y z radius r g b metalness roughness (host.random_spheres, a --spheres file) as add_spheres() takes them."""
    out = []
    for x, y, z, radius, r, g, b, metalness, roughness in np.asarray(rows9, np.float32):
        m = np.zeros((), dtype=scenes.MATERIAL_DTYPE)
        m["base_color"] = [r, g, b, 1.0]
        m["metalness"] = metalness
        m["roughness"] = roughness
        out.append(((x, y, z), radius, m))
    return out


def spheres_only_bundle(rows9, width, height, camera, **meta_kw):
    """A scene without a single triangle: every array the kernels are handed holds 48 zero bytes (one unused record), the threaded
    tree is one node that holds nothing, the flattened layout has sphere leaves only."""
    nodes = np.zeros((1, 8), np.uint32)
    nodes[0, 3] = nodes[0, 7] = 0xFFFFFFFF
    cam = host.camera(camera["pos"], camera["target"], camera["v_fov"], width / height, camera["aperture"], camera["focus"])
    base = dict(kind="spheres", positions=np.zeros(12, np.float32), normals=np.zeros(12, np.float32), uvs=np.zeros(12, np.float32),
                indices=np.zeros(12, np.uint32), materials=np.zeros(1, dtype=scenes.MATERIAL_DTYPE), textures=[None],
                triangles=np.zeros((0, 9), np.float32), ranks=None, camera=camera, nodes=nodes,
                meta=scenes.pack_metadata(cam, width, height, **meta_kw), width=width, height=height)
    return add_spheres(base, sphere_tuples(rows9))


def oracle_frames(bundle, spp, use_flat=False, materials=None, textures=None, aov=False, threads=None, first_sample=1, spheres=None):
    """Renders sample_index = first_sample .. first_sample+spp-1 with the CPU oracle, ping-ponging the two images
    like src/main.cpp:406-416.  Returns (image, counters of the last frame, (aov_prim, aov_t) of the last frame)."""
    import pyoracle
    arrays = dict(bundle)
    if materials is not None:
        arrays["materials"] = materials
    if textures is not None:
        arrays["textures"] = textures
    if spheres is not None:
        arrays["sphere_records"] = spheres
    o = pyoracle.OracleScene(arrays)
    H, W = bundle["height"], bundle["width"]
    acc = np.zeros((H, W, 4), np.float32)
    out = np.zeros_like(acc)
    counters, cap = None, None
    for s in range(first_sample, first_sample + spp):
        o.set_meta(sample_index=s)
        if aov:
            cap = (np.full((H, W, 2), -1, np.int32), np.zeros((H, W, 2), np.float32))
        counters = o.render(acc, out, use_flat=use_flat, aov=cap, threads=threads)
        acc, out = out, acc
    return acc, counters, cap


def gpu_frames(ctx, bundle, spp, pipeline, gpu_scene=None, capture=False, stats=False):
    """Same protocol on the GPU through the C ABI.  Returns (image, GpuScene, capture array or None, stats or None)."""
    from path_tracer_amd import hip
    W, H = bundle["width"], bundle["height"]
    gs = gpu_scene or hip.GpuScene(ctx, bundle, bundle["nodes"], bundle["flat"], bundle["meta"])
    gs.set_meta(bundle["meta"])
    loop = hip.FrameLoop(ctx, gs, W, H, pipeline)
    cap_buf = None
    if capture:
        cap_buf = ctx.create_buffer(W * H * 16)
        ctx.set_capture(cap_buf)
    if stats:
        ctx.stats_read(reset=True)
    for s in range(spp):
        if stats and s == spp - 1:
            ctx.stats_enable(True)
        loop.frame()
    img = loop.latest().read()
    st = None
    if stats:
        st = ctx.stats_read(reset=True)
        ctx.stats_enable(False)
    cap = None
    if capture:
        raw = cap_buf.read(np.uint8).reshape(H, W, 16)
        cap = (np.ascontiguousarray(raw[..., [0, 1, 2, 3, 8, 9, 10, 11]]).view(np.int32).reshape(H, W, 2),
               np.ascontiguousarray(raw[..., [4, 5, 6, 7, 12, 13, 14, 15]]).view(np.float32).reshape(H, W, 2))
        ctx.set_capture(None)
        cap_buf.destroy()
    return img, gs, cap, st


# ---- flattened-layout checker (host builder and GPU builder emit the same format) ----
def parse(flat):
    h = host.flat_header(flat)
    nodes = flat[h["nodes_offset"]:h["nodes_offset"] + 64 * h["node_count"]].view(np.float32).reshape(-1, 16)
    tris = flat[h["triangles_offset"]:h["triangles_offset"] + 64 * h["triangle_count"]].view(np.float32).reshape(-1, 16)
    return h, nodes, tris


def check_flat_layout(flat, triangles, ranks, tight=True):
    """Structural invariants of a flattened layout over `triangles` ((N, 9) float32); `ranks` = expected tie-break ranks.
    tight=False drops the two-step tightness bound (a builder may quantise more loosely, never less conservatively)."""
    b = dict(flat=flat, triangles=np.asarray(triangles, np.float32).reshape(-1, 9), ranks=np.asarray(ranks))
    h, nodes, tris = parse(b["flat"])
    n = b["triangles"].shape[0]
    assert h["magic"] == 0x46525450 and h["version"] == 2 and h["triangle_count"] == n
    assert h["nodes_offset"] % 64 == 0 and h["triangles_offset"] % 64 == 0 and b["flat"].nbytes >= 64 + 64 * (h["node_count"] + n)
    ti = tris.view(np.int32)
    assert sorted(ti[:, 7].tolist()) == list(range(n))                        # every primitive exactly once
    assert np.array_equal(ti[:, 3], b["ranks"][ti[:, 7]])                      # rank travels with the primitive
    tri = b["triangles"].reshape(n, 3, 3)[ti[:, 7]]
    assert np.array_equal(tris[:, 0:3], tri[:, 0])                             # p0, e1, e2, n = cross(e1, e2) in fp32
    e1, e2 = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
    assert np.array_equal(tris[:, 4:7], e1) and np.array_equal(tris[:, 8:11], e2)
    nrm = np.stack([e1[:, 1] * e2[:, 2] - e2[:, 1] * e1[:, 2], e1[:, 2] * e2[:, 0] - e2[:, 2] * e1[:, 0],
                    e1[:, 0] * e2[:, 1] - e2[:, 0] * e1[:, 1]], 1)
    assert np.array_equal(tris[:, 12:15], nrm)

    # walk the 4-wide tree: every decoded child box contains its subtree (conservative quantisation, within one
    # step of the true bound), leaves hold 1..4 triangles and tile the triangle array, the stack bound holds
    ni = nodes.view(np.int32)
    nu = nodes.view(np.uint32)
    seen = np.zeros(n, bool)
    tmin, tmax = tri.min(axis=1), tri.max(axis=1)
    EMPTY = -2 ** 31
    worst_stack = 0

    def bounds_of(ref, pending):
        nonlocal worst_stack
        if ref < 0:
            first, cnt = (~ref) >> 2, ((~ref) & 3) + 1
            assert not seen[first:first + cnt].any()
            seen[first:first + cnt] = True
            return tmin[first:first + cnt].min(axis=0), tmax[first:first + cnt].max(axis=0)
        nd = nodes[ref]
        count = sum(1 for k in range(4) if int(ni[ref, 4 + k]) != EMPTY)
        assert 2 <= count <= 4
        step = np.array([nd[3], nd[14], nd[15]], dtype=np.float32)
        assert all(float(x) > 0 and np.frexp(x)[0] == 0.5 for x in step), "steps are powers of two"
        worst_stack = max(worst_stack, pending + count - 1)
        lo_all, hi_all = [], []
        for k in range(4):
            child = int(ni[ref, 4 + k])
            if k >= count:
                assert child == EMPTY
                continue
            assert child != EMPTY
            qlo = np.array([(int(nu[ref, 8 + a]) >> (8 * k)) & 0xFF for a in range(3)], dtype=np.float32)
            qhi = np.array([(int(nu[ref, 11 + a]) >> (8 * k)) & 0xFF for a in range(3)], dtype=np.float32)  # words 11, 12, 13
            dlo = nd[0:3] + qlo * step
            dhi = nd[0:3] + qhi * step
            lo, hi = bounds_of(child, pending + count - 1)
            assert (dlo <= lo).all() and (dhi >= hi).all(), "decoded box must contain the subtree"
            if tight:
                assert (lo - dlo <= step * 2.001 + 1e-30).all() and (dhi - hi <= step * 2.001 + 1e-30).all(), "and be tight to two steps"
            lo_all.append(lo); hi_all.append(hi)
        lo, hi = np.min(lo_all, axis=0), np.max(hi_all, axis=0)
        slack = (hi - lo) * 2e-6 + 2 * np.spacing(np.abs(lo).astype(np.float32)) + 1e-29
        assert (nd[0:3] < lo).all() and (lo - nd[0:3] <= slack).all(), "node origin sits a margin below the true box minimum"
        return lo, hi

    import sys
    sys.setrecursionlimit(10000)
    lo, hi = bounds_of(h["root"], 0)
    assert seen.all() and worst_stack <= h["max_depth"] <= 254
    assert np.array_equal(lo, h["scene_min"]) and np.array_equal(hi, h["scene_max"])
    return h


