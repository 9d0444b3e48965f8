"""Known-answer tests of the CPU oracle's shader arithmetic.  The reference holds no vectors for this
path (no tests, no golden images of the current shader), so these values are hand-derived from the
shader text: parity of the shader arithmetic is pinned by them and by the restatement only."""
import ctypes as C

import numpy as np
import pytest

import pyoracle

L = pyoracle.lib()


def f32(*a):
    return np.array(a, dtype=np.float32)


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)  # keeps the array alive for the duration of the call


# ---- rand.h --------------------------------------------------------------------------------------------
def py_pcg(state):
    """rand.h:1-7 in integer arithmetic.  `seed *= 747796405u + 2891336453u`: the sum (mod 2^32) multiplies."""
    mult = (747796405 + 2891336453) & 0xFFFFFFFF
    new = (state * mult) & 0xFFFFFFFF
    word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & 0xFFFFFFFF
    return new, ((word >> 22) ^ word) & 0xFFFFFFFF


def test_rng_multiplier_is_even_and_state_dies_after_32_draws():
    assert (747796405 + 2891336453) & 0xFFFFFFFF == 3639132858
    for start in (1, 0xDEADBEEF | 1, (7 * 1973 + 3 * 9277 + 26699) | 1):
        s = C.c_uint32(start)
        py = start
        for i in range(40):
            got = L.orc_pcg_hash(C.byref(s))
            py, want = py_pcg(py)
            assert got == want and s.value == py
            if i >= 31:
                assert s.value == 0
        assert L.orc_rand(C.byref(s)) == 0.0  # hash(0) = 0


def test_rand_is_float_of_u32_over_2_pow_32_and_can_reach_one():
    s = C.c_uint32(1)
    _, h = py_pcg(1)
    assert h == 277803737 ^ (277803737 >> 22)
    assert L.orc_rand(C.byref(s)) == np.float32(h) / np.float32(4294967296.0)
    assert np.float32(0xFFFFFFFF) / np.float32(4294967296.0) == np.float32(1.0)


# ---- color_utils.h -------------------------------------------------------------------------------------
def test_srgb_curves():
    out = f32(0, 0, 0)
    L.orc_srgb_to_linear(ptr(f32(0.0, 1.0, 0.04)), ptr(out))
    assert out[0] == 0.0 and abs(out[1] - 1.0) < 2e-6
    assert out[2] == np.float32(0.04) / np.float32(12.92)           # linear toe below 0.04045
    L.orc_srgb_to_linear(ptr(f32(0.5, -3.0, 7.0)), ptr(out))
    assert abs(out[0] - ((0.5 + 0.055) / 1.055) ** 2.4) < 1e-6
    assert out[1] == 0.0 and abs(out[2] - 1.0) < 2e-6               # clamped to [0,1] first
    L.orc_linear_to_srgb(ptr(f32(0.002, 0.5, 1.0)), ptr(out))
    assert out[0] == np.float32(0.002) * np.float32(12.92)
    assert abs(out[1] - (1.055 * 0.5 ** (1 / 2.4) - 0.055)) < 1e-6
    assert abs(out[2] - 1.0) < 2e-6
    # round trip over the whole range
    x = np.linspace(0, 1, 257, dtype=np.float32)
    for v in x:
        a = f32(v, v, v); b = f32(0, 0, 0); c = f32(0, 0, 0)
        L.orc_linear_to_srgb(ptr(a), ptr(b)); L.orc_srgb_to_linear(ptr(b), ptr(c))
        assert abs(c[0] - v) < 3e-6


def test_polynomial_transcendentals_are_accurate():
    xs = np.linspace(0, 2 * np.pi, 4001, dtype=np.float32)
    s = np.array([L.orc_sin(float(x)) for x in xs]); c = np.array([L.orc_cos(float(x)) for x in xs])
    assert np.abs(s - np.sin(xs.astype(np.float64))).max() < 3e-7
    assert np.abs(c - np.cos(xs.astype(np.float64))).max() < 3e-7
    assert L.orc_sin(0.0) == 0.0 and L.orc_cos(0.0) == 1.0
    for base in np.geomspace(1e-30, 1.06, 200, dtype=np.float32):
        for e in (2.4, 1 / 2.4):
            got = L.orc_pow(float(base), float(np.float32(e)))
            want = float(base) ** float(np.float32(e))
            # the fp32 rounding of y*log2(x) bounds the accuracy: relative error ~ eps * |y*log2(x)|
            # (the sRGB curves only use bases >= 0.0031308, where this is < 1e-6)
            assert abs(got - want) <= 1e-7 * (3 + abs(e * np.log2(float(base)))) * want + 1e-44
    assert L.orc_pow(0.0, 2.4) == 0.0 and L.orc_pow(1.0, 2.4) == 1.0


# ---- ray.h ---------------------------------------------------------------------------------------------
def hit_tri(p0, p1, p2, o, d, tmin=0.001, tmax=1e15):
    t = C.c_float(); b = f32(0, 0, 0); g = f32(0, 0, 0)
    h = L.orc_hit_triangle_pos(ptr(f32(*p0)), ptr(f32(*p1)), ptr(f32(*p2)), ptr(f32(*o)), ptr(f32(*d)), tmin, tmax,
                               C.byref(t), ptr(b), ptr(g))
    return h, t.value, b, g


def test_unit_triangle_hits():
    tri = ((0, 0, 0), (1, 0, 0), (0, 1, 0))
    h, t, b, g = hit_tri(*tri, (0.25, 0.25, 1), (0, 0, -1))
    assert h == 1 and t == 1.0 and np.array_equal(b, f32(0.5, 0.25, 0.25)) and np.array_equal(g, f32(0, 0, 1))
    # no back-face culling, unnormalised direction scales t
    h, t, b, g = hit_tri(*tri, (0.25, 0.25, -2), (0, 0, 4))
    assert h == 1 and t == 0.5 and np.array_equal(g, f32(0, 0, 1))
    # edges are inclusive (u = 0 passes `u < 0`), outside rejects
    assert hit_tri(*tri, (0.0, 0.5, 1), (0, 0, -1))[0] == 1
    assert hit_tri(*tri, (0.5, 0.5, 1), (0, 0, -1))[0] == 1
    assert hit_tri(*tri, (0.51, 0.5, 1), (0, 0, -1))[0] == 0
    assert hit_tri(*tri, (-0.01, 0.5, 1), (0, 0, -1))[0] == 0
    # t range: t < min_t and t > max_t reject, t == max_t accepts
    assert hit_tri(*tri, (0.25, 0.25, 1), (0, 0, -1), 0.001, 0.5)[0] == 0
    assert hit_tri(*tri, (0.25, 0.25, 1), (0, 0, -1), 0.001, 1.0)[0] == 1
    assert hit_tri(*tri, (0.25, 0.25, 1), (0, 0, 1))[0] == 0  # behind the origin
    # a fully degenerate triangle yields NaN everywhere and NaN passes every rejection test (ray.h:153)
    h, t, _, _ = hit_tri((1, 1, 1), (1, 1, 1), (1, 1, 1), (0, 0, 0), (0, 0, 1))
    assert h == 1 and np.isnan(t)


def test_aabb_slab_test_and_nan_policy():
    lo, hi = f32(0, 0, 0), f32(1, 1, 1)
    hit = lambda o, d, a=0.001, b=1e15: L.orc_hit_aabb(ptr(lo), ptr(hi), ptr(f32(*o)), ptr(f32(*d)), a, b)
    assert hit((0.5, 0.5, -1), (0, 0, 1)) == 1
    assert hit((0.5, 0.5, -1), (0, 0, -1)) == 0
    assert hit((0.5, 0.5, 0.5), (1, 0, 0)) == 1           # origin inside
    assert hit((0.5, 0.5, -1), (0, 0, 1), 0.001, 0.5) == 0  # box beyond max_t
    assert hit((2, 0.5, -1), (0, 0, 1)) == 0
    # direction component 0 with the origin ON a slab plane: (0 - 0) * inf = NaN.  minimumNumber/maximumNumber
    # ignore the NaN, so tmin.x = min(inf, NaN) = +inf (origin on the min plane) or tmax.x = max(NaN, -inf) = -inf
    # (origin on the max plane): the declared policy makes both a MISS (GLSL leaves them undefined)
    assert hit((0.0, 0.5, -1), (0, 0, 1)) == 0
    assert hit((1.0, 0.5, -1), (0, 0, 1)) == 0
    assert hit((0.5, 0.5, -1), (0, 0, 1)) == 1              # strictly inside the slab: +-inf bracket, hit


def test_sphere_roots():
    t = C.c_float()
    hs = lambda o, d, a=0.001, b=1e15: L.orc_hit_sphere(ptr(f32(0, 0, 0)), 1.0, ptr(f32(*o)), ptr(f32(*d)), a, b, C.byref(t))
    assert hs((0, 0, -3), (0, 0, 1)) == 1 and t.value == 2.0    # nearer root
    assert hs((0, 0, 0), (0, 0, 1)) == 1 and t.value == 1.0     # inside: nearer root is negative -> farther root
    assert hs((0, 0, -3), (0, 0, 2)) == 1 and t.value == 1.0    # unnormalised direction
    assert hs((0, 2, -3), (0, 0, 1)) == 0                       # discriminant < 0
    assert hs((0, 0, -3), (0, 0, 1), 0.001, 1.5) == 0           # both roots beyond max_t
    assert hs((0, 1, -3), (0, 0, 1)) == 1 and t.value == 3.0    # tangent: discriminant == 0


# ---- math.h / brdf.h -------------------------------------------------------------------------------------
def test_rotation_to_z_axis():
    q = f32(0, 0, 0, 0)
    L.orc_rotation_to_z_axis(ptr(f32(0, 0, -1)), ptr(q))
    assert np.array_equal(q, f32(1, 0, 0, 0))                    # special case
    L.orc_rotation_to_z_axis(ptr(f32(0, 0, 1)), ptr(q))
    assert np.array_equal(q, f32(0, 0, 0, 1))                    # identity
    out = f32(0, 0, 0)
    for axis in ([1, 0, 0], [0, 1, 0], [0.6, 0.0, 0.8], [0.36, -0.48, 0.8]):
        a = f32(*axis)
        L.orc_rotation_to_z_axis(ptr(a), ptr(q))
        L.orc_rotate_point(ptr(q), ptr(a), ptr(out))
        assert np.abs(out - f32(0, 0, 1)).max() < 3e-7
        qi = f32(-q[0], -q[1], -q[2], q[3]); back = f32(0, 0, 0)
        L.orc_rotate_point(ptr(qi), ptr(out), ptr(back))
        assert np.abs(back - a).max() < 5e-7


def test_hemisphere_and_disk_samples():
    out = f32(0, 0, 0)
    L.orc_sample_hemisphere(0.0, 0.3, ptr(out)); assert np.array_equal(out, f32(0, 0, 1))
    L.orc_sample_hemisphere(1.0, 0.0, ptr(out)); assert np.array_equal(out, f32(1, 0, 0))
    L.orc_sample_hemisphere(0.25, 0.25, ptr(out))
    assert abs(out[0]) < 1e-7 and abs(out[1] - 0.5) < 1e-7 and out[2] == np.sqrt(np.float32(0.75))
    d = f32(0, 0)
    L.orc_disk_vec(0.5, 0.0, ptr(d)); assert np.array_equal(d, f32(1, 0))   # z = 0 -> r = 1: a projected sphere, not a uniform disk
    L.orc_disk_vec(1.0, 0.7, ptr(d)); assert np.array_equal(d, f32(0, 0))   # z = 1 -> r = 0


def test_brdf_probability_clamps_and_ggx_sample_is_unit_upper_hemisphere():
    view, n = f32(0, 0, 1), f32(0, 0, 1)
    assert L.orc_brdf_probability(ptr(f32(1, 1, 1)), 1.0, ptr(view), ptr(n)) == np.float32(0.9)   # pure metal -> specular, clamped
    assert L.orc_brdf_probability(ptr(f32(1, 1, 1)), 0.0, ptr(view), ptr(n)) == np.float32(0.1)   # white dielectric at normal incidence
    p = L.orc_brdf_probability(ptr(f32(0.5, 0.5, 0.5)), 0.5, ptr(view), ptr(n))
    f0 = 0.04 * 0.5 + 0.5 * 0.5
    assert abs(p - f0 / (f0 + 0.25 * (1 - f0))) < 1e-6
    out = f32(0, 0, 0)
    for seed in (1, 12345, 0x80000001):
        L.orc_sample_ggx_vndf(ptr(f32(0.3, -0.2, 0.9327379)), 0.25, 0.25, seed, ptr(out))
        assert abs(np.linalg.norm(out.astype(np.float64)) - 1) < 1e-6 and out[2] >= 0


def test_bilinear_repeat_texture_fetch():
    tex = np.array([[[0, 0, 0, 255], [255, 0, 0, 255]], [[0, 255, 0, 255], [255, 255, 255, 255]]], dtype=np.uint8)
    t = pyoracle.Texture(tex.ctypes.data, 2, 2)
    out = f32(0, 0, 0, 0)
    s = lambda u, v: (L.orc_sample_texture(C.byref(t), u, v, ptr(out)), out.copy())[1]
    assert np.array_equal(s(0.25, 0.25), f32(0, 0, 0, 1))          # texel centres
    assert np.array_equal(s(0.75, 0.25), f32(1, 0, 0, 1))
    assert np.array_equal(s(0.25, 0.75), f32(0, 1, 0, 1))
    assert np.array_equal(s(0.5, 0.25), f32(0.5, 0, 0, 1))         # halfway in x
    assert np.array_equal(s(0.5, 0.5), f32(0.5, 0.5, 0.25, 1))     # centre of the four
    assert np.array_equal(s(0.0, 0.25), f32(0.5, 0, 0, 1))         # REPEAT: blends with the opposite edge
    assert np.array_equal(s(1.25, -0.75), s(0.25, 0.25))           # periodic
    assert np.array_equal(s(float("nan"), 0.25), s(0.0, 0.25))     # NaN coordinate -> texel 0


def test_camera_ray_matches_hand_evaluation():
    from path_tracer_amd import host, scenes
    cam = host.camera([13, 2, -3], [0, 0, 0], 90, 16 / 9, 0.1, 10)
    meta = scenes.pack_metadata(cam, 400, 225)
    o = f32(0, 0, 0); d = f32(0, 0, 0)
    seed = (10 * 1973 + 20 * 9277 + 1 * 26699) | 1
    L.orc_camera_ray(ptr(meta), 10, 20, seed, ptr(o), ptr(d))
    assert np.array_equal(o, f32(13, 2, -3))                       # depth of field off: no lens offset
    m = meta.view(np.float32)
    s = seed
    draws = []
    for _ in range(4):
        s, h = py_pcg(s)
        draws.append(np.float32(h) / np.float32(4294967296.0))
    jx = np.float32(10) / np.float32(399) + draws[2] / np.float32(399)
    jy = (np.float32(1) - np.float32(20) / np.float32(224)) + draws[3] / np.float32(224)
    proj = (m[24:27] + m[16:19] * jx) + m[20:23] * jy              # first_pixel + jx*horizontal + jy*vertical
    assert np.array_equal(d, (proj - m[0:3]) - f32(0, 0, 0))
    # depth of field on: the shader's `right`/`up` are the C++ struct's up/right (compute.comp:48-50 vs camera.hpp:26-28)
    meta2 = scenes.pack_metadata(cam, 400, 225, enable_dof=1)
    L.orc_camera_ray(ptr(meta2), 10, 20, seed, ptr(o), ptr(d))
    disk = f32(0, 0); L.orc_disk_vec(float(draws[0]), float(draws[1]), ptr(disk))
    lens = disk * m[28]
    off = m[12:15] * lens[0] + m[8:11] * lens[1]                   # bytes 48.. (C++ up) * x + bytes 32.. (C++ right) * y
    assert np.array_equal(o, m[0:3] + off)
