"""Two-level scenes on the GPU: the wavefront kernels over a top level of placed copies against the CPU restatement of
the same traversal -- images, the captured (copy, primitive) pairs and hit distances, and the traversal counters, bit
for bit.  (Instancing extends the reference, whose scene is one triangle soup: src/scene.cpp:84-176.)"""
import numpy as np
import pytest

from common import gpu_frames, make_bundle, oracle_frames
from test_two_level import assorted_transforms, two_level_bundle
from path_tracer_amd import hip, host, scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = hip.Context(0)
    yield c
    c.close()


def oracle_for(bundle, gs, spp, **kw):
    return oracle_frames(bundle, spp, materials=gs.materials_host, textures=gs.texture_table, **kw)


def same(a, b, what):
    if not np.array_equal(a, b):
        bad = np.abs(a - b).max(axis=-1) > 0 if a.ndim == 3 else a != b
        raise AssertionError(f"{what}: {int(bad.sum())} of {bad.size} entries differ (first at {np.argwhere(bad)[0].tolist()})")


@pytest.mark.parametrize("which,kind,target,bounces", [("grid", "tiny", 0, 3), ("assorted", "tiny", 0, 6), ("assorted", "atrium", 6000, 4)])
def test_two_level_is_bit_exact_vs_oracle(ctx, which, kind, target, bounces):
    base = make_bundle(kind, target, 32, 24)
    m = scenes.grid_transforms(base, 3, 2) if which == "grid" else assorted_transforms(base, 6)
    two, _ = two_level_bundle(base, m, 112, 72, max_bounce=bounces)
    img, gs, cap, st = gpu_frames(ctx, two, 3, "compute.wavefront", capture=True, stats=True)
    ref, counters, aov = oracle_for(two, gs, 3, use_flat=True, aov=True)
    assert (aov[0][..., 0] >= 0).mean() > 0.05, "the camera sees the copies"
    assert len(np.unique(aov[0][..., 0][aov[0][..., 0] >= 0] // base["triangles"].shape[0])) >= 4, "several copies are hit"
    same(cap[0], aov[0], "copy * triangles + primitive of the first two hits")
    same(cap[1], aov[1], "hit distances of the first two hits")
    same(img, ref, "image")
    assert {k: st[k] for k in ("rays", "nodes", "triangles", "shaded")} == {k: counters[k] for k in ("rays", "nodes", "triangles", "shaded")}


def test_two_level_with_frames_in_flight_and_rebinding(ctx):
    """The overlapped frame slots render the same images; unbinding the top level gives the object alone again."""
    base = make_bundle("tiny", 0, 64, 40, max_bounce=4)
    two, _ = two_level_bundle(base, assorted_transforms(base, 5), 64, 40, max_bounce=4)
    img, gs, _, _ = gpu_frames(ctx, two, 5, "compute.wavefront")
    ref, _, _ = oracle_for(two, gs, 5, use_flat=True)
    same(img, ref, "five frames, two in flight")
    ctx.bind_instances(gs.flat_buffer, None)
    alone = dict(base, meta=two["meta"])
    img1, _, _, _ = gpu_frames(ctx, alone, 2, "compute.wavefront", gpu_scene=gs)
    ref1, _, _ = oracle_for(alone, gs, 2, use_flat=True)
    same(img1, ref1, "after unbinding")


def test_single_copy_tree_and_deep_stack(ctx):
    """One copy: the top level's root is the instance itself.  A stack bound near the limit sizes the spill area."""
    base = make_bundle("tiny", 0, 48, 32, max_bounce=3)
    m = assorted_transforms(base, 1)
    two, _ = two_level_bundle(base, m, 48, 32, max_bounce=3)
    assert host.flat_header(two["flat_instances"])["root"] < 0
    img, gs, cap, _ = gpu_frames(ctx, two, 2, "compute.wavefront", capture=True)
    ref, _, aov = oracle_for(two, gs, 2, use_flat=True, aov=True)
    same(cap[0], aov[0], "ids"); same(img, ref, "image")


def test_bad_instance_blobs_are_refused(ctx):
    base = make_bundle("tiny", 0, 32, 24, max_bounce=2)
    m = scenes.grid_transforms(base, 2, 2)
    good = host.bvh_flatten_instances(base["flat"], m)
    h = host.flat_header(good)

    def attempt(blob):
        two = dict(base, flat_instances=blob)
        gs = hip.GpuScene(ctx, two, two["nodes"], two["flat"], two["meta"])
        loop = hip.FrameLoop(ctx, gs, 32, 24, "compute.wavefront")
        loop.frame()

    bad = good.copy(); bad[:4] = base["flat"][:4]                                  # a triangle layout where the top level belongs
    with pytest.raises(hip.HipError, match="magic"):
        attempt(bad)
    bad = good.copy(); bad[36:40].view(np.uint32)[0] = 1                            # stack bound below the object's own
    with pytest.raises(hip.HipError, match="stack bound"):
        attempt(bad)
    rec = h["triangles_offset"]
    bad = good.copy(); bad[rec + 64 + 48:rec + 64 + 52].view(np.int32)[0] = 10 ** 6   # an instance whose object root is no node
    with pytest.raises(hip.HipError, match="instances"):
        attempt(bad)
    nodes = h["nodes_offset"]
    bad = good.copy(); bad[nodes + 16:nodes + 20].view(np.int32)[0] = ~np.int32(0x40000000 | 77)   # a leaf naming copy 77 of 4
    with pytest.raises(hip.HipError, match="instances"):
        attempt(bad)
    attempt(good)


def test_copies_of_two_objects_are_bit_exact_vs_oracle(ctx):
    """Two different objects in one layout (bvh::flatten_objects), seven copies naming them in turn."""
    from test_two_level import two_object_bundle
    b = two_object_bundle(112, 72, max_bounce=5)
    img, gs, cap, st = gpu_frames(ctx, b, 3, "compute.wavefront", capture=True, stats=True)
    ref, counters, aov = oracle_for(b, gs, 3, use_flat=True, aov=True)
    same(cap[0], aov[0], "copy * triangles + primitive of the first two hits")
    same(cap[1], aov[1], "hit distances of the first two hits")
    same(img, ref, "image")
    assert {k: st[k] for k in ("rays", "nodes", "triangles", "shaded")} == {k: counters[k] for k in ("rays", "nodes", "triangles", "shaded")}


@pytest.mark.parametrize("turns", [0, 2, 1000000])
def test_last_ray_of_a_wave_in_a_two_level_scene(ctx, turns):
    """extend.alone_turns with placed copies: the wave takes over a ray that may be inside a copy (marker on its stack, ray in
    object space); every pending entry is tagged with its copy and the hit, copy included, comes out the same."""
    from test_two_level import two_object_bundle
    b = two_object_bundle(128, 80, max_bounce=5)
    ctx.set_option("extend.alone_turns", turns)
    try:
        img, gs, cap, _ = gpu_frames(ctx, b, 3, "compute.wavefront", capture=True)
        img2, _, _, _ = gpu_frames(ctx, b, 4, "compute.wavefront", gpu_scene=gs)
    finally:
        ctx.set_option("extend.alone_turns", 0)
    ref, _, aov = oracle_for(b, gs, 3, use_flat=True, aov=True)
    same(cap[0], aov[0], "copy * triangles + primitive of the first two hits")
    same(cap[1], aov[1], "hit distances")
    same(img, ref, f"image, alone_turns {turns}")
    ref2, _, _ = oracle_for(b, gs, 4, use_flat=True)
    same(img2, ref2, f"image of the overlapped frames, alone_turns {turns}")


def test_wide_frontier_inside_placed_copies(ctx):
    """Three copies of a 341-node tree that every ray must visit in full: the takeover's list (references and their copies)
    outgrows the half-sized LDS ring of two-level scenes, so blocks of both arrays move out to the spill area and back."""
    from test_gpu_parity import fan_layout
    base = make_bundle("tiny", 0, 24, 16, max_bounce=2)
    base = dict(base, flat=fan_layout(base, 5))
    two, _ = two_level_bundle(base, assorted_transforms(base, 3), 40, 24, max_bounce=2)
    img, gs, cap, _ = gpu_frames(ctx, two, 2, "compute.wavefront", capture=True)
    ref, _, aov = oracle_for(two, gs, 2, use_flat=True, aov=True)
    assert len(np.unique(aov[0][..., 0])) > 8, "many different triangles are in view"
    same(cap[0], aov[0], "copy * triangles + primitive of the first two hits")
    same(cap[1], aov[1], "hit distances")
    same(img, ref, "image")


@pytest.mark.parametrize("turns", [0, 1000000])
@pytest.mark.parametrize("bounces", [1, 2, 4])
def test_two_level_last_bounce_stops_at_the_first_hit(ctx, bounces, turns):
    """Without the capture the traversal of the last bounce ends at the first hit, also inside a copy and in the cooperative
    finish of a wave's last ray: the image stays the oracle's."""
    base = make_bundle("atrium", 4000, 32, 24)
    two, _ = two_level_bundle(base, assorted_transforms(base, 6), 96, 60, max_bounce=bounces, min_bounce=0)
    ctx.set_option("extend.alone_turns", turns)
    try:
        img, gs, _, _ = gpu_frames(ctx, two, 3, "compute.wavefront")
    finally:
        ctx.set_option("extend.alone_turns", 0)
    ref, _, _ = oracle_for(two, gs, 3, use_flat=True)
    same(img, ref, f"{bounces} bounces")
