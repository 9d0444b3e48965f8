"""Sphere leaves on the GPU against the CPU oracle, through the C ABI (ptr_bvh_bind_spheres): whole images, primitive ids and
hit distances compared for equality, as for triangles."""
import numpy as np
import pytest

from common import add_spheres, gpu_frames, make_bundle, oracle_frames
from test_spheres import SPHERE_TAG, material, sphere_set

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from path_tracer_amd import hip
    c = hip.Context(0)
    yield c
    c.close()


def same(a, b, what):
    if not np.array_equal(a, b):
        bad = np.abs(a - b).max(axis=-1) > 0 if a.ndim == 3 and a.dtype.kind == "f" else a != b
        raise AssertionError(f"{what}: {int(bad.sum())} of {bad.size} differ (first at {np.argwhere(bad)[0].tolist()})")


def many_spheres(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        kind = rng.integers(0, 3)
        m = material(0.3 + 0.7 * rng.random(3), *[(0.0, 1.0), (1.0, 0.0), (1.0, float(rng.uniform(0.2, 0.9)))][kind])
        out.append((tuple(rng.uniform([-3.5, 0.2, -3.5], [3.5, 4.0, 3.5])), float(rng.uniform(0.05, 0.6)), m))
    return out


@pytest.mark.parametrize("turns", [0, 1000000])
@pytest.mark.parametrize("spheres,w,h,bounces", [("three", 64, 40, 5), ("many", 96, 54, 6)])
def test_spheres_beside_triangles_are_bit_exact_vs_oracle(ctx, spheres, w, h, bounces, turns):
    b = add_spheres(make_bundle("tiny", 0, w, h, max_bounce=bounces), sphere_set() if spheres == "three" else many_spheres(200, 3))
    ctx.set_option("extend.alone_turns", turns)   # 0: every wave's last ray is finished by all 64 lanes, sphere leaves included
    try:
        img, gs, cap, st = gpu_frames(ctx, b, 3, "compute.wavefront", capture=True, stats=True)
    finally:
        ctx.set_option("extend.alone_turns", 0)
    ref, counters, aov = oracle_frames(b, 3, use_flat=True, aov=True, materials=gs.materials_host, textures=gs.texture_table,
                                       spheres=gs.spheres_host)
    assert counters["nan_hits"] == 0
    on_sphere = (aov[0] >= 0) & ((aov[0] & SPHERE_TAG) != 0)
    assert on_sphere[..., 0].any() and on_sphere[..., 1].any(), "spheres are seen directly and after a bounce"
    same(cap[0], aov[0], "primitive ids")
    same(cap[1], aov[1], "hit distances")
    same(img, ref, "image")
    if turns == 1000000:   # the all-lanes finish visits another set of nodes; the single-lane walk is the oracle's
        assert {k: st[k] for k in ("rays", "shaded")} == {k: counters[k] for k in ("rays", "shaded")}


def test_spheres_without_any_triangle_in_view(ctx):
    """An atrium scene whose camera sits inside a mirror sphere next to a diffuse one: paths that only ever meet spheres."""
    b = make_bundle("tiny", 0, 40, 24, max_bounce=8)
    cam = tuple(np.float32(b["camera"]["pos"]))
    b = add_spheres(b, [(cam, 1.5, material((0.9, 0.9, 0.9), 1.0, 0.0)), ((cam[0] - 0.6, cam[1] - 0.3, cam[2] - 0.6), 0.25, material((0.9, 0.2, 0.2), 0.0, 1.0))])
    img, gs, cap, _ = gpu_frames(ctx, b, 2, "compute.wavefront", capture=True)
    ref, _, aov = oracle_frames(b, 2, use_flat=True, aov=True, materials=gs.materials_host, textures=gs.texture_table, spheres=gs.spheres_host)
    assert ((aov[0] & SPHERE_TAG) != 0).all()
    same(cap[0], aov[0], "primitive ids")
    same(cap[1], aov[1], "hit distances")
    same(img, ref, "image")


def test_sphere_leaves_need_their_records(ctx):
    from path_tracer_amd import hip
    b = add_spheres(make_bundle("tiny", 0, 32, 20), sphere_set())
    gs = hip.GpuScene(ctx, b, b["nodes"], b["flat"], b["meta"])
    loop = hip.FrameLoop(ctx, gs, 32, 20, "compute.wavefront")
    loop.frame()
    ctx.sync()
    # fewer records than the leaves name, or none at all: the dispatch is refused by the contents check, nothing faults
    ctx.bind_spheres(gs.flat_buffer, gs.sphere_buffer, 2)
    with pytest.raises(hip.HipError, match="flattened layout"):
        loop.frame()
        ctx.sync()
    ctx.bind_spheres(gs.flat_buffer, None)
    with pytest.raises(hip.HipError, match="flattened layout"):
        loop.frame()
        ctx.sync()
    ctx.bind_spheres(gs.flat_buffer, gs.sphere_buffer, 3)
    loop.frame()
    ctx.sync()
    with pytest.raises(hip.HipError, match="fewer than"):
        ctx.bind_spheres(gs.flat_buffer, gs.sphere_buffer, 4)
    with pytest.raises(hip.HipError):
        ctx.bind_spheres(gs.flat_buffer, gs.flat_buffer, 1)
    # the megakernel walks the threaded nodes, which hold no spheres: it renders the triangles alone
    plain = make_bundle("tiny", 0, 32, 20)
    img, _, _, _ = gpu_frames(ctx, plain, 1, "compute.megakernel", gpu_scene=gs)
    ref, _, _ = oracle_frames(plain, 1, materials=gs.materials_host, textures=gs.texture_table)
    same(img, ref, "megakernel image")


def test_spheres_and_placed_copies_do_not_combine(ctx):
    from path_tracer_amd import hip, host
    b = add_spheres(make_bundle("tiny", 0, 32, 20), sphere_set())
    identity = np.array([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]]], np.float32)
    b["flat_instances"] = host.bvh_flatten_instances(b["flat"], identity)
    with pytest.raises(hip.HipError, match="sphere leaves cannot be instanced"):
        hip.GpuScene(ctx, b, b["nodes"], b["flat"], b["meta"])      # binds the instances first, then the spheres
    gs = hip.GpuScene(ctx, dict(b, flat_instances=None), b["nodes"], b["flat"], b["meta"])
    top = ctx.create_buffer(b["flat_instances"].nbytes)
    top.write(b["flat_instances"])
    with pytest.raises(hip.HipError, match="sphere leaves cannot be instanced"):
        ctx.bind_instances(gs.flat_buffer, top)
    hip.FrameLoop(ctx, gs, 32, 20, "compute.wavefront").frame()      # and the scene still renders
    ctx.sync()
