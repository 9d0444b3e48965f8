"""Sphere leaves on the CPU side: the layout the host builder writes for them and the oracle's treatment of them.
The reference keeps hit_sphere (shaders/include/ray.h:105-135) but its call site is commented out (ray.h:173-177), so no image
of the reference shows a sphere: the intersection routine is pinned by its formula, the frame of a hit by the commented lines
ray.h:127-132 (point on the ray, normal (point - centre) / radius); everything downstream is the triangle path."""
import ctypes as C

import numpy as np
import pytest

from common import add_spheres, make_bundle, oracle_frames, parse
from path_tracer_amd import host, scenes

SPHERE_TAG = 0x40000000


def material(base, metal, rough):
    m = np.zeros((), dtype=scenes.MATERIAL_DTYPE)
    m["base_color"] = list(base) + [1.0]
    m["metalness"] = metal
    m["roughness"] = rough
    return m


def sphere_set():
    # a diffuse ball on the floor, a mirror ball overlapping the triangulated one, a small rough-metal ball near the camera
    return [((-1.5, 0.8, 1.5), 0.8, material((0.2, 0.6, 0.9), 0.0, 1.0)),
            ((0.9, 1.4, 0.6), 0.9, material((1.0, 1.0, 1.0), 1.0, 0.0)),
            ((1.8, 1.9, 2.4), 0.3, material((0.9, 0.7, 0.2), 1.0, 0.4))]


def test_layout_holds_sphere_records_beside_the_triangles():
    b = add_spheres(make_bundle("tiny", 0, 32, 20), sphere_set())
    h, nodes, recs = parse(b["flat"])
    n = b["triangles"].shape[0]
    assert h["triangle_count"] == n + 3
    ri = recs.view(np.int32)
    kinds = ri[:, 11]
    assert sorted(kinds.tolist()) == [0] * n + [1] * 3
    sph = recs[kinds == 1]
    ids = ri[kinds == 1][:, 7]
    assert sorted(ids.tolist()) == [SPHERE_TAG | i for i in range(3)]
    for rec, pid in zip(sph, ids):
        centre, radius, _ = sphere_set()[pid & ~SPHERE_TAG]
        assert np.array_equal(rec[0:3], np.float32(centre)) and rec[4] == np.float32(radius)
    assert sorted(ri[kinds == 0][:, 7].tolist()) == list(range(n))            # the triangles are all still there, once each
    # every sphere's box lies inside the root's decoded child boxes' union: walk down from the root to its leaf
    ni, nu = nodes.view(np.int32), nodes.view(np.uint32)

    def leaf_box(ref, lo, hi, want):
        if ref < 0:
            first, cnt = (~ref) >> 2, ((~ref) & 3) + 1
            return (lo, hi) if first <= want < first + cnt else None
        nd = nodes[ref]
        step = np.array([nd[3], nd[14], nd[15]], np.float32)
        for k in range(4):
            child = int(ni[ref, 4 + k])
            if child == -2 ** 31:
                continue
            qlo = np.array([(int(nu[ref, 8 + a]) >> (8 * k)) & 0xFF for a in range(3)], np.float32)
            qhi = np.array([(int(nu[ref, 11 + a]) >> (8 * k)) & 0xFF for a in range(3)], np.float32)
            got = leaf_box(child, nd[0:3] + qlo * step, nd[0:3] + qhi * step, want)
            if got is not None:
                return (np.maximum(lo, got[0]), np.minimum(hi, got[1]))      # the tightest box on the way down
        return None
    for slot in np.flatnonzero(kinds == 1):
        centre, radius = recs[slot, 0:3], recs[slot, 4]
        lo, hi = leaf_box(h["root"], np.full(3, -np.inf, np.float32), np.full(3, np.inf, np.float32), int(slot))
        assert (lo <= centre - radius).all() and (hi >= centre + radius).all(), "every box above a sphere contains it"


def test_spheres_only_layout():
    geometry = np.array([[0, 0, 0, 1], [3, 0, 0, 0.5]], np.float32)
    flat = host.bvh_flatten(np.zeros((0, 9), np.float32), None, spheres=geometry)
    h, _, recs = parse(flat)
    assert h["triangle_count"] == 2 and sorted(recs.view(np.int32)[:, 11].tolist()) == [1, 1]


def test_hit_sphere_follows_the_shader_formula():
    """ray.h:105-135 in fp32: a = |d|^2, half_b = oc.d, c = |oc|^2 - r^2, the nearer root first, then the farther one."""
    import pyoracle
    L = pyoracle.lib()
    f32 = np.float32
    rng = np.random.default_rng(5)
    fp = C.POINTER(C.c_float)

    def arr(v):
        return np.ascontiguousarray(v, dtype=np.float32)
    hits = 0
    for _ in range(2000):
        centre = arr(rng.normal(size=3) * 2)
        radius = f32(rng.uniform(0.2, 1.5))
        origin = arr(rng.normal(size=3) * 3)
        d = arr((centre + rng.normal(size=3) * 0.8 * radius - origin) * rng.uniform(0.2, 2.0))   # aimed near the sphere, not unit length
        min_t, max_t = f32(0.001), f32(rng.uniform(1, 20))
        t = C.c_float(0)
        got = L.orc_hit_sphere(centre.ctypes.data_as(fp), radius, origin.ctypes.data_as(fp), d.ctypes.data_as(fp), min_t, max_t, C.byref(t))
        oc = origin - centre
        # fp64 restatement with generous margins: only clear cases are asserted
        a, hb, c = float(d @ d), float(oc.astype(np.float64) @ d), float(oc.astype(np.float64) @ oc) - float(radius) ** 2
        disc = hb * hb - a * c
        if disc < -1e-3:
            assert got == 0
            continue
        if disc < 1e-3:
            continue
        roots = [(-hb - disc ** 0.5) / a, (-hb + disc ** 0.5) / a]
        inside = [r for r in roots if float(min_t) + 1e-3 < r < float(max_t) - 1e-3]
        near_edge = any(abs(r - float(min_t)) < 2e-3 or abs(r - float(max_t)) < 2e-3 for r in roots)
        if near_edge:
            continue
        if inside:
            assert got == 1 and abs(t.value - inside[0]) <= 1e-3 * max(1.0, abs(inside[0])), (t.value, roots)
            hits += 1
        else:
            assert got == 0
    assert hits > 100


def test_oracle_sees_a_sphere_that_encloses_the_camera_from_inside():
    b = make_bundle("tiny", 0, 24, 16, max_bounce=2)
    cam = np.float32(b["camera"]["pos"])
    b = add_spheres(b, [(tuple(cam), 0.5, material((0.5, 0.5, 0.5), 0.0, 1.0))])
    _, counters, aov = oracle_frames(b, 1, use_flat=True, aov=True)
    first = aov[0][..., 0]
    assert (first == (SPHERE_TAG | 0)).all(), "every primary ray ends on the inside of the sphere around the lens"
    assert np.all(aov[1][..., 0] > 0)      # in units of the primary direction, which is not of unit length (camera.h:18-27)


def test_oracle_image_with_spheres_differs_only_where_a_sphere_can_be_seen_or_lit():
    plain = make_bundle("tiny", 0, 48, 30, max_bounce=4)
    with_spheres = add_spheres(plain, sphere_set())
    ref, _, aov0 = oracle_frames(plain, 2, use_flat=True, aov=True)
    img, counters, aov = oracle_frames(with_spheres, 2, use_flat=True, aov=True)
    assert counters["nan_hits"] == 0 and np.isfinite(img).all()
    on_sphere = (aov[0][..., 0] >= 0) & ((aov[0][..., 0] & SPHERE_TAG) != 0)      # a miss is -1
    assert on_sphere.any() and not on_sphere.all()
    seen = set((aov[0][..., 0][on_sphere] & ~SPHERE_TAG).tolist())
    assert seen == {0, 1, 2}, seen
    # a primary hit on a sphere is never farther than what the ray hit without the spheres
    t_with, t_plain = aov[1][..., 0], aov0[1][..., 0]
    hit_plain = aov0[0][..., 0] >= 0
    assert (t_with[on_sphere & hit_plain] <= t_plain[on_sphere & hit_plain]).all()
    # and a pixel whose two first hits avoid the spheres starts the same way in both scenes
    same_start = ~on_sphere
    assert np.array_equal(aov[0][..., 0][same_start], aov0[0][..., 0][same_start])
    assert not np.array_equal(img, ref)


def test_sphere_count_limits_of_the_host_builder():
    with pytest.raises(host.HostError):
        host.bvh_flatten(np.zeros((0, 9), np.float32), None, spheres=np.array([[0, 0, 0, -1.0]], np.float32))
    with pytest.raises(host.HostError):
        host.bvh_flatten(np.zeros((0, 9), np.float32), None, spheres=np.array([[0, np.nan, 0, 1.0]], np.float32))
