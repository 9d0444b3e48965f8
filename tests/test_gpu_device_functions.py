"""The device code at FUNCTION granularity: every shader function the render kernels are composed of is evaluated
on the GPU through the C ABI (ptr_device_function) and compared, bit for bit, with the CPU oracle's unit entry
point on the same seeded argument records, plus the hand-derived known answers the oracle itself is pinned by.
Covers hit_sphere (ray.h:105-135), which no frame exercises because its call site in the reference is commented out."""
import ctypes as C

import numpy as np
import pytest

import pyoracle

pytestmark = pytest.mark.gpu

L = pyoracle.lib()
N = 3000


def fptr(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def ctx():
    from path_tracer_amd import hip
    c = hip.Context(0)
    yield c
    c.close()


def special_floats():
    return np.array([0.0, -0.0, 1.0, -1.0, 0.5, 2.0, 1e-30, 1e30, 3.0e-39, np.inf, -np.inf, np.nan, 0.04045, 0.0031308,
                     0.001, 1e15, 255.0, 1.0 / 3.0], np.float32)


def random_records(rng, n, words, scale=1.0):
    a = (rng.standard_normal((n, words)) * scale).astype(np.float32)
    sp = special_floats()
    k = n // 5
    rows = rng.integers(0, n, k)
    cols = rng.integers(0, words, k)
    a[rows, cols] = sp[rng.integers(0, len(sp), k)]   # sprinkle zeros, infinities, NaN, denormals
    return a


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def assert_same_bits(got, want, what, floats=True):
    """Bit equality; for float words any NaN equals any NaN (x86 and the GPU differ in the sign of a generated NaN)."""
    got = np.array(got, np.uint32)
    want = np.array(want, np.uint32)
    if floats:
        for a in (got, want):
            a[(a & 0x7FFFFFFF) > 0x7F800000] = 0x7FC00000
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, f"{what}: {bad.size} of {got.shape[0]} records differ, first #{bad[0]}: gpu {got[bad[0]]} cpu {want[bad[0]]}"


def test_rng(ctx):
    rng = np.random.default_rng(1)
    seeds = np.concatenate([rng.integers(0, 2**32, N, dtype=np.uint64).astype(np.uint32), np.array([0, 1, 0xFFFFFFFF, 0x80000000], np.uint32)])
    got_h = ctx.device_function("pcg_hash", seeds[:, None], 2)
    got_r = ctx.device_function("rand", seeds[:, None], 2)
    want_h = np.zeros_like(got_h)
    want_r = np.zeros_like(got_r)
    for i, s in enumerate(seeds):
        st = C.c_uint32(int(s))
        want_h[i, 0] = L.orc_pcg_hash(C.byref(st)); want_h[i, 1] = st.value
        st = C.c_uint32(int(s))
        want_r[i, 0] = bits(np.float32(L.orc_rand(C.byref(st))).reshape(1))[0]; want_r[i, 1] = st.value
    assert_same_bits(got_h, want_h, "pcg_hash", floats=False)
    assert_same_bits(got_r, want_r, "rand")
    # the even multiplier kills the state: 32 chained draws end in 0
    s = np.array([[0xDEADBEEF | 1]], np.uint32)
    for _ in range(32):
        s = ctx.device_function("pcg_hash", s, 2)[:, 1:2]
    assert s[0, 0] == 0


@pytest.mark.parametrize("name,orc", [("sin", "orc_sin"), ("cos", "orc_cos")])
def test_sin_cos(ctx, name, orc):
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(-7, 7, N), rng.uniform(-1e4, 1e4, N // 4), special_floats(),
                        np.arange(0, 65) * (2 * np.pi / 64)]).astype(np.float32)
    # the argument reduction converts x * 2/pi to int32: specified for finite |x| < 2^31 * pi/2 only (the shader
    # passes 2*pi*r with r in [0, 1])
    x = x[np.isfinite(x) & (np.abs(x) <= 1e4)]
    got = ctx.device_function(name, bits(x)[:, None], 1)
    f = getattr(L, orc)
    want = bits(np.array([f(float(v)) for v in x], np.float32))[:, None]
    assert_same_bits(got, want, name)


def test_pow(ctx):
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(0, 4, N), special_floats()]).astype(np.float32)
    y = np.concatenate([rng.choice([2.4, 1 / 2.4, 5.0, 0.5, 2.0], N), rng.uniform(-3, 3, len(special_floats()))]).astype(np.float32)
    rec = np.stack([x, y], 1)
    got = ctx.device_function("pow", bits(rec), 1)
    want = bits(np.array([L.orc_pow(float(a), float(b)) for a, b in rec], np.float32))[:, None]
    assert_same_bits(got, want, "pow")


@pytest.mark.parametrize("name", ["srgb_to_linear", "linear_to_srgb"])
def test_colour_curves(ctx, name):
    rng = np.random.default_rng(4)
    rec = np.concatenate([rng.uniform(-0.2, 1.2, (N, 3)), np.resize(special_floats(), (6, 3))]).astype(np.float32)
    got = ctx.device_function(name, bits(rec), 3)
    f = getattr(L, "orc_" + name)
    want = np.zeros_like(rec)
    for i in range(rec.shape[0]):
        f(fptr(rec[i]), fptr(want[i]))
    assert_same_bits(got, bits(want), name)
    if name == "srgb_to_linear":    # color_utils.h: clamp, linear segment below 0.04045
        k = ctx.device_function(name, bits(np.array([[0.0, 1.0, 0.04], [0.5, -3.0, 7.0]], np.float32)), 3).view(np.float32)
        assert k[0, 0] == 0.0 and k[0, 1] == 1.0 and k[0, 2] == np.float32(0.04) / np.float32(12.92)
        assert k[1, 1] == 0.0 and k[1, 2] == 1.0


def test_hit_aabb(ctx):
    rng = np.random.default_rng(5)
    lo = rng.uniform(-2, 0, (N, 3)); hi = lo + rng.uniform(0, 2, (N, 3))
    o = rng.uniform(-4, 4, (N, 3)); d = rng.standard_normal((N, 3))
    aim = rng.random(N) < 0.6                                                 # most rays aimed near the box
    d[aim] = (lo + (hi - lo) * rng.uniform(-0.3, 1.3, (N, 3)) - o)[aim]
    d[rng.integers(0, N, N // 6), rng.integers(0, 3, N // 6)] = 0.0          # axis-parallel rays (0 * inf)
    on = rng.integers(0, N, N // 10)
    o[on, 0] = lo[on, 0]                                                      # origin on a slab plane
    tr = np.stack([np.full(N, 0.001), rng.choice([1e15, 1.0, 3.0], N)], 1)
    rec = np.concatenate([lo, hi, o, d, tr], 1).astype(np.float32)
    got = ctx.device_function("hit_aabb", bits(rec), 1)
    want = np.array([[L.orc_hit_aabb(fptr(r[0:3]), fptr(r[3:6]), fptr(r[6:9]), fptr(r[9:12]), float(r[12]), float(r[13]))] for r in rec], np.uint32)
    assert_same_bits(got, want, "hit_aabb")
    assert 0.15 < want.mean() < 0.95
    # ray.h:66-78 known answers: through the box, away from it, starting inside, box beyond max_t
    unit = [-1, -1, -1, 1, 1, 1]
    k = np.array([unit + [0, 0, -3, 0, 0, 1, 0.001, 1e15], unit + [0, 0, -3, 0, 0, -1, 0.001, 1e15],
                  unit + [0, 0, 0, 0.3, 0.2, 1, 0.001, 1e15], unit + [0, 0, -3, 0, 0, 1, 0.001, 1.5]], np.float32)
    assert ctx.device_function("hit_aabb", bits(k), 1)[:, 0].tolist() == [1, 0, 1, 0]


def test_hit_triangle(ctx):
    rng = np.random.default_rng(6)
    p = rng.uniform(-1, 1, (N, 9))
    p[:N // 50, 3:6] = p[:N // 50, 0:3]                                       # degenerate: two equal corners -> NaN t accepted
    o = rng.uniform(-1, 1, (N, 3)) + np.array([0, 0, -3.0]); d = rng.standard_normal((N, 3)) * 0.3 + np.array([0, 0, 1.0])
    w = rng.uniform(-0.2, 0.8, (N, 3, 1)); w /= w.sum(axis=1, keepdims=True)  # most rays aimed at / just beside the triangle
    aim = rng.random(N) < 0.7
    d[aim] = ((p.reshape(N, 3, 3) * w).sum(axis=1) - o)[aim] * rng.uniform(0.3, 2.0, (N, 1))[aim]
    tr = np.stack([np.full(N, 0.001), rng.choice([1e15, 3.0], N)], 1)
    rec = np.concatenate([p, o, d, tr], 1).astype(np.float32)
    got = ctx.device_function("hit_triangle", bits(rec), 8)
    want = np.zeros((N, 8), np.uint32)
    t = C.c_float(); bary = np.zeros(3, np.float32); gn = np.zeros(3, np.float32)
    for i, r in enumerate(rec):
        hit = L.orc_hit_triangle_pos(fptr(r[0:3]), fptr(r[3:6]), fptr(r[6:9]), fptr(r[9:12]), fptr(r[12:15]), float(r[15]), float(r[16]),
                                     C.byref(t), fptr(bary), fptr(gn))
        want[i, 0] = hit
        if hit:
            want[i, 1] = bits(np.float32(t.value).reshape(1))[0]; want[i, 2:5] = bits(bary); want[i, 5:8] = bits(gn)
    hits = want[:, 0] == 1
    assert np.array_equal(got[:, 0], want[:, 0])
    assert_same_bits(got[hits], want[hits], "hit_triangle")
    assert 0.05 < hits.mean() < 0.9
    # ray.h:137-163 known answer: unit right triangle in z = 0, hit at (0.25, 0.25): t 3, bary (0.5, 0.25, 0.25), normal +z
    k = np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0, 0.25, 0.25, -3, 0, 0, 1, 0.001, 1e15]], np.float32)
    r = ctx.device_function("hit_triangle", bits(k), 8)
    assert r[0, 0] == 1 and np.array_equal(r[0, 1:].view(np.float32), np.array([3, 0.5, 0.25, 0.25, 0, 0, 1], np.float32))


def test_hit_sphere(ctx):
    rng = np.random.default_rng(7)
    c = rng.uniform(-1, 1, (N, 3)); rad = rng.uniform(0.1, 1.5, (N, 1))
    o = rng.uniform(-3, 3, (N, 3)); d = rng.standard_normal((N, 3))
    aim = rng.random(N) < 0.7
    d[aim] = (c + rad * rng.uniform(-1.2, 1.2, (N, 3)) - o)[aim] * rng.uniform(0.3, 2.0, (N, 1))[aim]
    tr = np.stack([np.full(N, 0.001), rng.choice([1e15, 2.0], N)], 1)
    rec = np.concatenate([c, rad, o, d, tr], 1).astype(np.float32)
    got = ctx.device_function("hit_sphere", bits(rec), 2)
    want = np.zeros((N, 2), np.uint32)
    t = C.c_float()
    for i, r in enumerate(rec):
        hit = L.orc_hit_sphere(fptr(r[0:3]), float(r[3]), fptr(r[4:7]), fptr(r[7:10]), float(r[10]), float(r[11]), C.byref(t))
        want[i, 0] = hit
        if hit:
            want[i, 1] = bits(np.float32(t.value).reshape(1))[0]
    assert_same_bits(got, want, "hit_sphere")
    assert 0.05 < want[:, 0].mean() < 0.9
    # ray.h:105-135 known answers on the unit sphere: nearer root, inside -> farther root, unnormalised direction,
    # miss, both roots beyond max_t, tangent
    s = [0, 0, 0, 1]
    k = np.array([s + [0, 0, -3, 0, 0, 1, 0.001, 1e15], s + [0, 0, 0, 0, 0, 1, 0.001, 1e15], s + [0, 0, -3, 0, 0, 2, 0.001, 1e15],
                  s + [0, 2, -3, 0, 0, 1, 0.001, 1e15], s + [0, 0, -3, 0, 0, 1, 0.001, 1.5], s + [0, 1, -3, 0, 0, 1, 0.001, 1e15]], np.float32)
    r = ctx.device_function("hit_sphere", bits(k), 2)
    assert r[:, 0].tolist() == [1, 1, 1, 0, 0, 1]
    assert r[[0, 1, 2, 5], 1].view(np.float32).tolist() == [2.0, 1.0, 1.0, 3.0]


def unit_vectors(rng, n):
    v = rng.standard_normal((n, 3))
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def test_frames_and_sampling(ctx):
    rng = np.random.default_rng(8)
    axis = np.concatenate([unit_vectors(rng, N), np.array([[0, 0, -1], [0, 0, 1], [1, 0, 0]], np.float32)])
    got_q = ctx.device_function("rotation_to_z_axis", bits(axis), 4)
    want_q = np.zeros((axis.shape[0], 4), np.float32)
    for i in range(axis.shape[0]):
        L.orc_rotation_to_z_axis(fptr(axis[i]), fptr(want_q[i]))
    assert_same_bits(got_q, bits(want_q), "rotation_to_z_axis")
    assert np.array_equal(got_q[-3].view(np.float32), np.array([1, 0, 0, 0], np.float32))    # math.h:29-32 special case

    v = random_records(rng, axis.shape[0], 3)
    rec = np.concatenate([want_q, v], 1)
    got = ctx.device_function("rotate_point", bits(rec), 3)
    want = np.zeros((rec.shape[0], 3), np.float32)
    for i in range(rec.shape[0]):
        L.orc_rotate_point(fptr(rec[i, 0:4]), fptr(rec[i, 4:7]), fptr(want[i]))
    assert_same_bits(got, bits(want), "rotate_point")

    r = np.concatenate([rng.uniform(0, 1, (N, 2)), np.array([[0, 0.3], [1, 0], [0.25, 0.25], [1, 1], [0, 0]])]).astype(np.float32)
    for name, words in (("sample_hemisphere", 3), ("disk_vec", 2)):
        got = ctx.device_function(name, bits(r), words)
        want = np.zeros((r.shape[0], words), np.float32)
        f = getattr(L, "orc_" + name)
        for i in range(r.shape[0]):
            f(float(r[i, 0]), float(r[i, 1]), fptr(want[i]))
        assert_same_bits(got, bits(want), name)


def test_brdf_terms(ctx):
    rng = np.random.default_rng(9)
    ve = unit_vectors(rng, N); ve[:, 2] = np.abs(ve[:, 2])
    alpha = rng.choice([0.0, 1e-4, 0.04, 0.25, 1.0], (N, 2)).astype(np.float32)
    alpha[:, 1] = alpha[:, 0]
    seeds = rng.integers(0, 2**32, N, dtype=np.uint64).astype(np.uint32)
    seeds[:8] = 0                                                              # dead RNG state: both draws are 0
    rec = np.concatenate([bits(ve), bits(alpha), seeds[:, None]], 1)
    got = ctx.device_function("sample_ggx_vndf", rec, 3)
    want = np.zeros((N, 3), np.float32)
    for i in range(N):
        L.orc_sample_ggx_vndf(fptr(ve[i]), float(alpha[i, 0]), float(alpha[i, 1]), int(seeds[i]), fptr(want[i]))
    assert_same_bits(got, bits(want), "sample_ggx_vndf")

    colour = rng.uniform(0, 1, (N, 3)).astype(np.float32)
    metal = rng.choice([0.0, 1.0, 0.5, 0.03], (N, 1)).astype(np.float32)
    view = unit_vectors(rng, N); n = unit_vectors(rng, N)
    rec = np.concatenate([colour, metal, view, n], 1)
    got = ctx.device_function("brdf_probability", bits(rec), 1)
    want = np.array([L.orc_brdf_probability(fptr(r[0:3]), float(r[3]), fptr(r[4:7]), fptr(r[7:10])) for r in rec], np.float32)
    assert_same_bits(got, bits(want)[:, None], "brdf_probability")
    assert want.min() >= np.float32(0.1) and want.max() <= np.float32(0.9)     # brdf.h:98 clamp


def test_device_function_errors(ctx):
    from path_tracer_amd import hip
    with pytest.raises(hip.HipError):
        ctx.device_function("no_such_function", np.zeros((1, 1), np.uint32), 1)
    with pytest.raises(hip.HipError):
        ctx.device_function("hit_sphere", np.zeros((1, 11), np.uint32), 2)      # wrong record size
    assert ctx.device_function("sin", np.zeros((0, 1), np.uint32), 1).shape == (0, 1)
