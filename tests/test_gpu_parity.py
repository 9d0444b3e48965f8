"""Parity of the HIP path against the CPU oracle, through the C ABI.  Integer/ID outputs and -- because both
sides implement the same fully specified fp32 arithmetic -- whole images are compared for equality
(numpy `==`: +0 equals -0, NaN never occurs in these scenes)."""
import numpy as np
import pytest

from common import gpu_frames, make_bundle, oracle_frames, parse

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from path_tracer_amd import hip
    c = hip.Context(0)
    yield c
    c.close()


def oracle_for(bundle, gs, spp, **kw):
    return oracle_frames(bundle, spp, materials=gs.materials_host, textures=gs.texture_table, **kw)


def assert_same(a, b, what):
    if not np.array_equal(a, b):
        bad = np.abs(a - b).max(axis=-1) > 0 if a.ndim == 3 else a != b
        raise AssertionError(f"{what}: {int(bad.sum())} of {bad.size} pixels differ (first at {np.argwhere(bad)[0].tolist()})")


@pytest.mark.parametrize("kind,target,w,h,bounces", [("tiny", 0, 64, 40, 3), ("tiny", 0, 50, 37, 6), ("atrium", 5000, 96, 54, 4)])
def test_megakernel_is_bit_exact_vs_oracle(ctx, kind, target, w, h, bounces):
    b = make_bundle(kind, target, w, h, max_bounce=bounces)
    img, gs, cap, st = gpu_frames(ctx, b, 3, "compute.megakernel", capture=True, stats=True)
    ref, counters, aov = oracle_for(b, gs, 3, aov=True)
    assert_same(img, ref, "image")
    assert_same(cap[0], aov[0], "primitive ids of the first two hits")
    assert_same(cap[1], aov[1], "hit distances of the first two hits")
    # traversal work is identical too: node fetches, triangle tests, rays, shaded hits of the last frame
    assert {k: st[k] for k in ("rays", "nodes", "triangles", "shaded")} == {k: counters[k] for k in ("rays", "nodes", "triangles", "shaded")}


@pytest.mark.parametrize("kind,target,w,h,bounces", [("tiny", 0, 64, 40, 3), ("tiny", 0, 50, 37, 6), ("atrium", 5000, 96, 54, 4)])
def test_wavefront_is_bit_exact_vs_oracle_on_the_flattened_layout(ctx, kind, target, w, h, bounces):
    b = make_bundle(kind, target, w, h, max_bounce=bounces)
    img, gs, cap, st = gpu_frames(ctx, b, 3, "compute.wavefront", capture=True, stats=True)
    ref, counters, aov = oracle_for(b, gs, 3, use_flat=True, aov=True)
    assert_same(img, ref, "image")
    assert_same(cap[0], aov[0], "primitive ids")
    assert_same(cap[1], aov[1], "hit distances")
    assert {k: st[k] for k in ("rays", "nodes", "triangles", "shaded")} == {k: counters[k] for k in ("rays", "nodes", "triangles", "shaded")}


def test_wavefront_matches_the_reference_order_traversal_up_to_declared_ties(ctx):
    """Closest-hit IDs of the fast path against the reference's own traversal order: equal except where the
    reference's box tests cull a true hit by rounding (the fast path pads its boxes) -- counted, not hidden."""
    b = make_bundle("atrium", 20000, 320, 180, max_bounce=2)
    img, gs, cap, _ = gpu_frames(ctx, b, 1, "compute.wavefront", capture=True)
    ref, counters, aov = oracle_for(b, gs, 1, aov=True)
    assert counters["nan_hits"] == 0
    differing_ids = int((cap[0][..., 0] != aov[0][..., 0]).sum())
    assert differing_ids <= max(1, int(1e-5 * cap[0][..., 0].size) + 1), differing_ids
    same = cap[0][..., 0] == aov[0][..., 0]
    assert np.array_equal(cap[1][..., 0][same], aov[1][..., 0][same])


def test_pregathered_shading_records_change_nothing_and_follow_buffer_writes(ctx):
    """The wavefront shade step reads per-triangle records gathered once from the index / normal / uv / position buffers:
    same frames with the records on or off, and a later Buffer::write to one of the sources is picked up."""
    b = make_bundle("atrium", 4000, 80, 48, max_bounce=4)
    img_on, gs, _, _ = gpu_frames(ctx, b, 2, "compute.wavefront")
    ctx.set_option("shade.pregather", 0)
    try:
        img_off, _, _, _ = gpu_frames(ctx, b, 2, "compute.wavefront", gpu_scene=gs)
    finally:
        ctx.set_option("shade.pregather", 1)
    ref, _, _ = oracle_for(b, gs, 2, use_flat=True)
    assert_same(img_on, ref, "records on vs oracle")
    assert_same(img_off, ref, "records off vs oracle")

    # tilt every vertex normal and swap the uv axes, in place, after the records were built
    n = np.asarray(b["normals"], np.float32).reshape(-1, 3).copy()
    n = n + np.float32(0.35) * np.roll(n, 1, axis=1)
    n = (n / np.linalg.norm(n, axis=1, keepdims=True)).astype(np.float32)
    uv = np.ascontiguousarray(np.asarray(b["uvs"], np.float32).reshape(-1, 2)[:, ::-1])
    gs.normals_buffer.write(n.reshape(-1))
    gs.uvs_buffer.write(uv.reshape(-1))
    b2 = dict(b, normals=n.reshape(-1), uvs=uv.reshape(-1))
    img2, _, _, _ = gpu_frames(ctx, b2, 2, "compute.wavefront", gpu_scene=gs)
    ref2, _, _ = oracle_for(b2, gs, 2, use_flat=True)
    assert not np.array_equal(ref2, ref)
    assert_same(img2, ref2, "after rewriting normals and uvs")


def test_frames_in_flight_change_nothing(ctx):
    """Consecutive frames are traced on internal streams, several in flight: the images are those of one stream, also when
    the materials buffer is rewritten between two dispatches without any synchronisation by the caller."""
    b = make_bundle("atrium", 4000, 88, 56, max_bounce=4)
    gs = None
    images = {}
    for k in (1, 2, 3, 4):
        ctx.set_option("frames.in_flight", k)
        try:
            img, gs, _, _ = gpu_frames(ctx, b, 7, "compute.wavefront", gpu_scene=gs)
        finally:
            ctx.set_option("frames.in_flight", 2)
        images[k] = img
    ref, _, _ = oracle_for(b, gs, 7, use_flat=True)
    for k, img in images.items():
        assert_same(img, ref, f"{k} frame(s) in flight vs oracle")

    # frame 1..3 with the original materials, then every base colour halved, then frames 4..6: no sync in between
    from path_tracer_amd import hip
    dimmed = gs.materials_host.copy()
    dimmed["base_color"][:, :3] *= np.float32(0.5)
    results = {}
    for k in (1, 3):
        ctx.set_option("frames.in_flight", k)
        try:
            gs.materials_buffer.write(gs.materials_host)
            gs.set_meta(b["meta"])
            loop = hip.FrameLoop(ctx, gs, b["width"], b["height"], "compute.wavefront")
            for _ in range(3):
                loop.frame()
            gs.materials_buffer.write(dimmed)
            for _ in range(3):
                loop.frame()
            results[k] = loop.latest().read()
        finally:
            ctx.set_option("frames.in_flight", 2)
            gs.materials_buffer.write(gs.materials_host)
    assert_same(results[3], results[1], "materials rewritten between frames in flight")
    assert not np.array_equal(results[1], images[1])


def test_wide_index_layout_and_instanced_grid(ctx):
    """The 4-word index layout (scenes past 2^24 vertices) renders exactly what the reference's packed layout renders,
    in both pipelines, and a grid of translated copies built with it matches the oracle."""
    from path_tracer_amd import host, scenes
    b = make_bundle("atrium", 3000, 72, 48, max_bounce=4)
    wide = scenes.widen_indices(b)
    for pipeline, use_flat in (("compute.megakernel", False), ("compute.wavefront", True)):
        img3, gs3, _, _ = gpu_frames(ctx, b, 2, pipeline)
        img4, gs4, _, _ = gpu_frames(ctx, wide, 2, pipeline)
        assert_same(img4, img3, f"{pipeline}: wide vs packed index words")
        ref, _, _ = oracle_for(wide, gs4, 2, use_flat=use_flat)
        assert_same(img4, ref, f"{pipeline}: wide layout vs oracle")
    with pytest.raises(Exception):
        ctx.set_option("scene.index_words", 5)

    grid = scenes.instance_grid(b, 3, 2)
    assert grid["triangles"].shape[0] == 6 * b["triangles"].shape[0]
    nodes = host.bvh_build(grid["triangles"])
    ranks = host.bvh_visit_ranks(nodes, grid["triangles"].shape[0])
    g = dict(grid, nodes=nodes, ranks=ranks, flat=host.bvh_flatten(grid["triangles"], ranks), meta=b["meta"], width=b["width"], height=b["height"])
    img, gs, cap, st = gpu_frames(ctx, g, 2, "compute.wavefront", capture=True, stats=True)
    ref, counters, aov = oracle_for(g, gs, 2, use_flat=True, aov=True)
    assert_same(img, ref, "instanced grid")
    assert_same(cap[0], aov[0], "instanced grid primitive ids")
    assert st["nodes"] == counters["nodes"] and st["triangles"] == counters["triangles"]
    ctx.set_option("scene.index_words", 3)


def test_frames_in_flight_survive_stream_changes_and_teardown():
    """Frames in flight when the caller changes the context's stream, reads back, or destroys the context."""
    import ctypes
    from path_tracer_amd import hip
    b = make_bundle("atrium", 3000, 64, 40, max_bounce=4)
    c = hip.Context(0)
    runtime = ctypes.CDLL("libamdhip64.so")          # the HIP runtime the library itself is linked against
    side = ctypes.c_void_p()
    assert runtime.hipStreamCreate(ctypes.byref(side)) == 0
    try:
        gs = hip.GpuScene(c, b, b["nodes"], b["flat"], b["meta"])
        loop = hip.FrameLoop(c, gs, b["width"], b["height"], "compute.wavefront")
        for i in range(6):
            if i == 2:
                c.set_stream(side.value)             # frames 0, 1 are in flight on the old stream
            if i == 4:
                c.set_stream(None)
            loop.frame()
        img = loop.latest().read()
        ref, _, _ = oracle_for(b, gs, 6, use_flat=True)
        assert_same(img, ref, "six frames across two stream changes")
        for _ in range(5):
            loop.frame()                             # destroyed with frames in flight
    finally:
        c.close()
        runtime.hipStreamDestroy(side)


def test_bad_scene_contents_fail_the_dispatch_instead_of_faulting(ctx):
    """Vertex ids, material ids, node links and child references are indices into other buffers: an out-of-range one is
    reported by ptr_dispatch (checked on the GPU once per upload), and the scene works again once it is repaired."""
    from path_tracer_amd import hip
    b = make_bundle("atrium", 3000, 48, 32, max_bounce=3)
    good, gs, _, _ = gpu_frames(ctx, b, 1, "compute.wavefront")
    n_vertices = len(b["positions"]) // 3
    idx = np.asarray(b["indices"], np.uint32).copy()

    def expect_failure(pipeline, what):
        with pytest.raises(hip.HipError, match=what):
            gpu_frames(ctx, b, 1, pipeline, gpu_scene=gs)

    bad = idx.copy(); bad[7] = (bad[7] & np.uint32(0xFF000000)) | np.uint32(n_vertices)           # one vertex id past the end
    gs.indices_buffer.write(bad)
    expect_failure("compute.wavefront", "vertex")
    expect_failure("compute.megakernel", "vertex")
    bad = idx.copy(); bad[2::3] |= np.uint32(0xFF000000)                                             # material 255 and up
    gs.indices_buffer.write(bad)
    expect_failure("compute.wavefront", "material")
    gs.indices_buffer.write(idx)

    nodes = np.asarray(b["nodes"], np.uint32).copy()
    bad = nodes.copy(); bad[5, 3] = np.uint32(len(nodes) + 10)                                      # a skip link past the end
    gs.bvh_buffer.write(bad)
    expect_failure("compute.megakernel", "bvh nodes")
    # a skip link that points backwards (or at its own node) is in range, yet the stackless walk would never end
    for at, to in ((5, 5), (9, 2)):
        bad = nodes.copy(); bad[at, 3] = np.uint32(to)
        gs.bvh_buffer.write(bad)
        expect_failure("compute.megakernel", "bvh nodes")
    gs.bvh_buffer.write(nodes)

    flat = np.asarray(b["flat"], np.uint8).copy()
    h = flat[:64].view(np.uint32)
    bad = flat.copy()
    bad[64 + 16:64 + 20].view(np.int32)[0] = int(h[2]) + 5                                          # root's first child: no such node
    gs.flat_buffer.write(bad)
    expect_failure("compute.wavefront", "flattened layout")
    gs.flat_buffer.write(flat)

    again, _, _, _ = gpu_frames(ctx, b, 1, "compute.wavefront", gpu_scene=gs)
    assert_same(again, good, "after repairing the buffers")


def test_sample_partition_exchange(ctx):
    """Two ranks (two partitions of one context here), each keeping the running mean of its own global samples:
    every rank's frames equal the oracle seeded with that rank's sample numbers, and linearize -> sum -> resolve is
    the per-pixel arithmetic it says it is."""
    import ctypes as C
    import pyoracle
    from path_tracer_amd import hip
    world, frames = 2, 3
    b = make_bundle("atrium", 3000, 56, 40, max_bounce=4)
    W, H = b["width"], b["height"]
    gs = hip.GpuScene(ctx, b, b["nodes"], b["flat"], b["meta"])
    L = pyoracle.lib()
    linear = []
    try:
        for rank in range(world):
            ctx.set_sample_partition(rank, world)
            gs.set_meta(b["meta"])
            loop = hip.FrameLoop(ctx, gs, W, H, "compute.wavefront")
            for _ in range(frames):
                loop.frame()
            img = loop.latest().read()
            o = pyoracle.OracleScene(dict(b, materials=gs.materials_host, textures=gs.texture_table))
            acc = np.zeros((H, W, 4), np.float32)
            out = np.zeros_like(acc)
            for k in range(1, frames + 1):
                o.set_meta(sample_index=k)
                o.seed_sample_index = (k - 1) * world + rank + 1
                o.render(acc, out, use_flat=True)
                acc, out = out, acc
            assert_same(img, acc, f"rank {rank} of {world}")
            # linearize on the device: (srgb_to_linear(rgb), 1)
            rp = ctx.create_compute_renderpass()
            rp.set_pipeline("linearize")
            rp.set_ouput_texture(loop.output)
            rp.set_constant(60, loop.latest())
            rp.set_dispatch_size(1, 1, 1)
            rp.execute()
            lin = loop.output.read()
            want = np.ones_like(img)
            for y in range(H):
                for x in range(W):
                    L.orc_srgb_to_linear(img[y, x, :3].ctypes.data_as(C.c_void_p), want[y, x, :3].ctypes.data_as(C.c_void_p))
            assert_same(lin, want, "linearize")
            linear.append(lin)
        # what the all-reduce does, then resolve on the device: linear_to_srgb(rgb / w)
        total = linear[0] + linear[1]
        loop.output.update(total)
        rp.set_pipeline("resolve")
        rp.set_ouput_texture(loop.latest())
        rp.set_constant(60, loop.output)
        rp.execute()
        final = loop.latest().read()
        want = np.ones_like(final)
        mean = (total[..., :3] / total[..., 3:4]).astype(np.float32)
        for y in range(H):
            for x in range(W):
                L.orc_linear_to_srgb(mean[y, x].ctypes.data_as(C.c_void_p), want[y, x, :3].ctypes.data_as(C.c_void_p))
        assert_same(final, want, "resolve")
        assert (total[..., 3] == world).all()
    finally:
        ctx.set_sample_partition(0, 1)


def test_auto_pipeline_picks_wavefront_when_a_flat_layout_is_bound(ctx):
    b = make_bundle("tiny", 0, 48, 32, max_bounce=3)
    img_auto, gs, _, _ = gpu_frames(ctx, b, 2, "compute")
    img_wf, _, _, _ = gpu_frames(ctx, b, 2, "compute.wavefront", gpu_scene=gs)
    assert_same(img_auto, img_wf, "auto vs wavefront")
    prof_before = ctx.profile_read()
    ctx.profile_enable(True)
    gpu_frames(ctx, b, 1, "compute", gpu_scene=gs)
    names = ctx.profile_read()
    ctx.profile_enable(False)
    assert "wavefront.extend" in names and names["wavefront.extend"]["launches"] == 3 and "compute.megakernel" not in names, (prof_before, names)
    ctx.bind_flat(gs.bvh_buffer, None)
    ctx.profile_enable(True)
    img_mega, _, _, _ = gpu_frames(ctx, b, 2, "compute", gpu_scene=gs)
    names = ctx.profile_read()
    ctx.profile_enable(False)
    assert names.get("compute.megakernel", {}).get("launches") == 2
    ref, _, _ = oracle_for(b, gs, 2)
    assert_same(img_mega, ref, "auto without flat layout = megakernel")


@pytest.mark.parametrize("pipeline,use_flat", [("compute.megakernel", False), ("compute.wavefront", True)])
@pytest.mark.parametrize("meta_kw", [dict(enable_dof=1), dict(downscale_factor=2), dict(min_bounce=0, max_bounce=8),
                                     dict(max_bounce=1), dict(max_bounce=0), dict(max_bounce=20, min_bounce=18)])
def test_metadata_flags(ctx, pipeline, use_flat, meta_kw):
    kw = dict(max_bounce=3)
    kw.update(meta_kw)
    b = make_bundle("tiny", 0, 64, 48, **kw)
    img, gs, _, _ = gpu_frames(ctx, b, 2, pipeline)
    ref, _, _ = oracle_for(b, gs, 2, use_flat=use_flat)
    assert_same(img, ref, f"{pipeline} {meta_kw}")


@pytest.mark.parametrize("pipeline,use_flat", [("compute.megakernel", False), ("compute.wavefront", True)])
@pytest.mark.parametrize("w,h,meta_kw", [(1, 1, {}), (7, 3, {}), (9, 65, dict(enable_dof=1)), (50, 37, dict(downscale_factor=3)), (33, 20, dict(downscale_factor=5))])
def test_ragged_image_sizes(ctx, pipeline, use_flat, w, h, meta_kw):
    """Sizes that are no multiple of the 8x8 workgroup, down to one pixel, and downscale factors that do not divide them."""
    b = make_bundle("tiny", 0, w, h, max_bounce=3, **meta_kw)
    img, gs, _, _ = gpu_frames(ctx, b, 2, pipeline)
    ref, _, _ = oracle_for(b, gs, 2, use_flat=use_flat)
    assert_same(img, ref, f"{pipeline} {w}x{h} {meta_kw}")


def test_dispatch_smaller_and_larger_than_the_image(ctx):
    """The workgroup count is the caller's: fewer groups than the image leave the rest untouched, more are clipped
    (compute.comp:265 guards with the metadata's width and height)."""
    from path_tracer_amd import hip
    b = make_bundle("tiny", 0, 40, 32, max_bounce=3)
    full, gs, _, _ = gpu_frames(ctx, b, 1, "compute.wavefront")
    for pipeline in ("compute.wavefront", "compute.megakernel"):
        for gx, gy in ((2, 1), (64, 64)):
            gs.set_meta(b["meta"])
            loop = hip.FrameLoop(ctx, gs, 40, 32, pipeline)
            loop.renderpass.set_dispatch_size(gx, gy, 1)
            loop.frame()
            img = loop.latest().read()
            covered = np.zeros((32, 40), bool)
            covered[:min(32, gy * 8), :min(40, gx * 8)] = True
            want = full if pipeline == "compute.wavefront" else gpu_frames(ctx, b, 1, pipeline, gpu_scene=gs)[0]
            assert np.array_equal(img[covered], want[covered]), (pipeline, gx, gy)
            assert (img[~covered] == 0).all(), (pipeline, gx, gy)


@pytest.mark.parametrize("turns", [0, 3, 1000000])
@pytest.mark.parametrize("kind,target,w,h,bounces", [("tiny", 0, 64, 40, 6), ("atrium", 5000, 160, 90, 4)])
def test_last_ray_of_a_wave_finished_by_all_lanes(ctx, turns, kind, target, w, h, bounces):
    """extend.alone_turns: when a traversal wave is down to one ray, all 64 lanes take its pending nodes and leaves, in
    another order than the single lane would -- the closest hit (ties by rank) does not depend on the order.  0 hands
    every wave's last ray over at once, 1000000 never does."""
    b = make_bundle(kind, target, w, h, max_bounce=bounces)
    ctx.set_option("extend.alone_turns", turns)
    try:
        img, gs, cap, _ = gpu_frames(ctx, b, 3, "compute.wavefront", capture=True)
        img2, _, _, _ = gpu_frames(ctx, b, 4, "compute.wavefront", gpu_scene=gs)      # two frames in flight (no capture)
    finally:
        ctx.set_option("extend.alone_turns", 0)
    ref, _, aov = oracle_for(b, gs, 3, use_flat=True, aov=True)
    assert_same(cap[0], aov[0], "primitive ids")
    assert_same(cap[1], aov[1], "hit distances")
    assert_same(img, ref, f"image, alone_turns {turns}")
    ref2, _, _ = oracle_for(b, gs, 4, use_flat=True)
    assert_same(img2, ref2, f"image of the overlapped frames, alone_turns {turns}")


def chain_layout(bundle, levels, claimed_depth):
    """A hand-made flattened layout: a chain of `levels` nodes, each with the next node as its first child and three leaves that
    all hold the scene's largest triangle, every child box equal to the whole scene -- a ray that enters stacks three entries per level."""
    from path_tracer_amd import host
    h0, _, tris0 = parse(bundle["flat"])
    ti = tris0.view(np.int32)
    area = np.linalg.norm(tris0[:, 12:15], axis=1)               # |e1 x e2|
    tri = tris0[int(np.argmax(area))].copy()                     # the record of the largest triangle
    lo, hi = h0["scene_min"].astype(np.float32), h0["scene_max"].astype(np.float32)
    origin = lo - np.float32(1.0)
    step = np.float32(2.0) ** np.ceil(np.log2((hi - origin + 1.0) / 255.0)).astype(np.float32)
    nodes = np.zeros((levels, 16), np.float32)
    ni, nu = nodes.view(np.int32), nodes.view(np.uint32)
    leaf = ~np.int32(0)                                           # record 0 of this layout, one triangle
    for k in range(levels):
        nodes[k, 0:3] = origin; nodes[k, 3] = step[0]; nodes[k, 14] = step[1]; nodes[k, 15] = step[2]
        ni[k, 4:8] = [k + 1 if k + 1 < levels else leaf, leaf, leaf, leaf]
        nu[k, 8:11] = 0                                            # lower planes at the origin
        nu[k, 11:14] = 0xFFFFFFFF                                  # upper planes 255 steps away
    header = np.zeros(64, np.uint8)
    header.view(np.uint32)[0:4] = [0x46525450, 2, levels, 1]
    header.view(np.uint64)[2:4] = [64, 64 + 64 * levels]
    header.view(np.int32)[8] = 0
    header.view(np.uint32)[9] = claimed_depth
    header.view(np.float32)[10:13] = lo; header.view(np.float32)[13:16] = hi
    return np.concatenate([header, nodes.view(np.uint8).reshape(-1), tri.view(np.uint8).reshape(-1)])


def fan_layout(bundle, depth):
    """A complete 4-ary tree of `depth` levels of nodes whose every box is the whole scene: a ray that enters the scene box has to
    visit all of it, and a breadth-first walk's frontier gets 4^depth wide.  Every triangle of the scene hangs off exactly one
    leaf (spread evenly over the leaves; the others repeat the first record), so a reference lost on the way shows in the picture."""
    _, _, tris0 = parse(bundle["flat"])
    n = tris0.shape[0]
    chain = chain_layout(bundle, 1, 3 * depth)
    node = chain[64:128].view(np.float32).copy()
    inner = (4 ** depth - 1) // 3
    leaves = 4 ** depth
    assert leaves >= n
    stride = leaves // n
    nodes = np.tile(node, (inner, 1))
    ni = nodes.view(np.int32)
    for k in range(inner):
        for c in range(4):
            child = 4 * k + 1 + c
            if child < inner:
                ni[k, 4 + c] = child
            else:
                leaf = child - inner
                record = leaf // stride if leaf % stride == 0 and leaf // stride < n else 0
                ni[k, 4 + c] = ~np.int32(record << 2)
    header = chain[:64].copy()
    header.view(np.uint32)[2:4] = [inner, n]
    header.view(np.uint64)[3] = 64 + 64 * inner
    return np.concatenate([header, nodes.view(np.uint8).reshape(-1), tris0.view(np.uint8).reshape(-1)])


@pytest.mark.parametrize("depth", [5, 6])
def test_wide_frontiers_in_the_wave_takeover(ctx, depth):
    """341 / 1365 nodes that every ray must visit, 1024 / 4096 leaf references pending at once: the list of the wave that finishes
    a ray together outgrows its LDS ring, so blocks of it move out to the spill area and back (and at depth 6 even that
    fills up and the turns get narrower)."""
    b = make_bundle("tiny", 0, 24, 16, max_bounce=2)
    b = dict(b, flat=fan_layout(b, depth))
    img, gs, cap, _ = gpu_frames(ctx, b, 2, "compute.wavefront", capture=True)
    ref, _, aov = oracle_for(b, gs, 2, use_flat=True, aov=True)
    assert len(np.unique(aov[0][..., 0])) > 8, "many different triangles are in view"
    assert_same(cap[0], aov[0], "primitive ids")
    assert_same(cap[1], aov[1], "hit distances")
    assert_same(img, ref, "image")


def test_a_frontier_wider_than_ring_and_spill_area_loses_nothing(ctx):
    """Depth 7: 16 384 leaf references pending at once, four times what the wave's LDS ring (1024) and its spill area (3072)
    hold together.  The takeover then narrows its turns, down to one reference per turn (a depth-first walk, whose need the
    layout header bounds), instead of dropping what does not fit: every hit still equals the oracle's."""
    b = make_bundle("tiny", 0, 24, 16, max_bounce=2)
    b = dict(b, flat=fan_layout(b, 7))
    img, gs, cap, _ = gpu_frames(ctx, b, 2, "compute.wavefront", capture=True)
    ref, _, aov = oracle_for(b, gs, 2, use_flat=True, aov=True)
    assert len(np.unique(aov[0][..., 0])) > 8, "many different triangles are in view"
    assert_same(cap[0], aov[0], "primitive ids")
    assert_same(cap[1], aov[1], "hit distances")
    assert_same(img, ref, "image")
    ctx.sync()                                                      # and no fault was raised on the way


def test_deep_traversal_stacks(ctx):
    """42 levels x 3 pending entries: far past the 16 entries a lane keeps in LDS, so the spill area, the pops from it and the
    wave takeover's gather of a deep stack all run."""
    b = make_bundle("tiny", 0, 48, 32, max_bounce=3)
    b = dict(b, flat=chain_layout(b, 42, 130))
    img, gs, cap, _ = gpu_frames(ctx, b, 2, "compute.wavefront", capture=True)
    ref, _, aov = oracle_for(b, gs, 2, use_flat=True, aov=True)
    assert (aov[0][..., 0] >= 0).mean() > 0.02, "the triangle is in view"
    assert_same(cap[0], aov[0], "primitive ids")
    assert_same(img, ref, "image")


def test_layout_header_that_understates_its_stack_is_refused(ctx):
    """The kernels size their stacks from the header's bound, so the library measures the tree before the first dispatch:
    42 levels x 3 pending entries under a header that says 4 fail the dispatch; so does a node reached along two paths."""
    from path_tracer_amd import hip
    b = make_bundle("tiny", 0, 48, 32, max_bounce=3)

    def attempt(flat):
        scene = dict(b, flat=flat)
        gs = hip.GpuScene(ctx, scene, scene["nodes"], scene["flat"], scene["meta"])
        hip.FrameLoop(ctx, gs, 48, 32, "compute.wavefront").frame()

    with pytest.raises(hip.HipError, match="deeper traversal stack"):
        attempt(chain_layout(b, 42, 4))
    with pytest.raises(hip.HipError, match="deeper traversal stack"):
        attempt(chain_layout(b, 42, 125))                         # one short of the 126 it needs
    attempt(chain_layout(b, 42, 126))
    shared = chain_layout(b, 6, 130)
    shared[64 + 16:64 + 32].view(np.int32)[1:4] = 1               # node 0 names node 1 four times over
    shared[64 + 64 + 16:64 + 64 + 32].view(np.int32)[1:4] = 2     # ... and so on: the frontier outgrows the node array
    shared[64 + 128 + 16:64 + 128 + 32].view(np.int32)[1:4] = 3
    with pytest.raises(hip.HipError, match="not a tree"):
        attempt(shared)
    ok = make_bundle("tiny", 0, 48, 32, max_bounce=3)             # the context goes on working
    img2, gs2, _, _ = gpu_frames(ctx, ok, 2, "compute.wavefront")
    assert_same(img2, oracle_for(ok, gs2, 2, use_flat=True)[0], "next scene")


def test_debug_bvh_heatmap(ctx):
    """compute.comp:274-275 / ray.h:80-103: boxes of the threaded tree a primary ray enters, x 0.001, in red.  The counts are
    defined on that tree, so every dispatch name gives the same picture -- also "compute" with a flattened layout bound
    (what the scene class always does), which must not quietly render a normal frame."""
    from path_tracer_amd import hip
    b = make_bundle("atrium", 3000, 64, 40, debug_bvh=1)
    img, gs, _, _ = gpu_frames(ctx, b, 1, "compute.megakernel")
    ref, _, _ = oracle_for(b, gs, 1)
    assert_same(img, ref, "debug_bvh")
    assert img[..., 0].max() > 0 and img[..., 1].max() == 0
    for pipeline in ("compute", "compute.wavefront"):
        assert_same(gpu_frames(ctx, b, 1, pipeline, gpu_scene=gs)[0], ref, f"debug_bvh through {pipeline} with a flattened layout bound")
    normal = gpu_frames(ctx, dict(b, meta=make_bundle("atrium", 3000, 64, 40)["meta"]), 1, "compute", gpu_scene=gs)[0]
    assert not np.array_equal(normal, ref)                         # the flag is what made the difference
    # no threaded tree to count on (a placeholder node buffer, as the large-scene benchmark uses): an error, not a picture
    placeholder = np.zeros((1, 8), np.uint32); placeholder[0, 3] = 0xFFFFFFFF
    gs2 = hip.GpuScene(ctx, b, placeholder, b["flat"], b["meta"])
    with pytest.raises(hip.HipError, match="debug_bvh"):
        hip.FrameLoop(ctx, gs2, 64, 40, "compute").frame()


def test_tonemapping_kernel(ctx):
    from path_tracer_amd import hip
    import pyoracle
    b = make_bundle("tiny", 0, 40, 24)
    gs = hip.GpuScene(ctx, b, b["nodes"], b["flat"], b["meta"])
    rng = np.random.default_rng(3)
    src = rng.random((24, 40, 4), dtype=np.float32)
    acc = ctx.create_2d_texture(40, 24, hip.FORMAT_RGBA32_SFLOAT); acc.update(src)
    out = ctx.create_2d_texture(40, 24, hip.FORMAT_RGBA32_SFLOAT)
    rp = ctx.create_compute_renderpass()
    rp.set_pipeline("tonemapping")
    gs.wire(rp)
    rp.set_ouput_texture(out)
    rp.set_constant(60, acc)
    rp.set_dispatch_size(40 // 8 + 1, 24 // 8 + 1, 1)
    rp.execute()
    got = out.read()
    want = np.zeros_like(src)
    o = pyoracle.OracleScene(dict(b, materials=gs.materials_host, textures=gs.texture_table))
    o.tonemap(src, want, 40 // 8 + 1, 24 // 8 + 1)
    assert_same(got, want, "tonemapping")
    assert np.array_equal(got[..., :3], np.sqrt(src[..., :3])) and (got[..., 3] == 1).all()


def test_stripe_partition_union_is_the_whole_frame(ctx):
    """Two contexts standing in for two GPUs: each renders its stripes; merged rows equal the single-GPU frame."""
    from path_tracer_amd import hip
    b = make_bundle("atrium", 4000, 80, 72, max_bounce=3)
    full, gs, _, _ = gpu_frames(ctx, b, 3, "compute.wavefront")
    for pipeline in ("compute.wavefront", "compute.megakernel"):
        whole = full if pipeline == "compute.wavefront" else gpu_frames(ctx, b, 3, pipeline, gpu_scene=gs)[0]
        parts = []
        for rank in range(2):
            ctx.set_partition(rank, 2, 8)
            img, _, _, _ = gpu_frames(ctx, b, 3, pipeline, gpu_scene=gs)
            parts.append(img)
        ctx.set_partition(0, 1, 8)
        rows = (np.arange(72) // 8) % 2
        merged = np.where((rows == 0)[:, None, None], parts[0], parts[1])
        assert_same(merged, whole, f"{pipeline}: union of stripes")
        assert (parts[0][rows == 1] == 0).all() and (parts[1][rows == 0] == 0).all()   # a rank never writes foreign rows


def test_error_paths(ctx):
    from path_tracer_amd import hip
    b = make_bundle("tiny", 0, 32, 16)
    gs = hip.GpuScene(ctx, b, b["nodes"], None, b["meta"])
    loop = hip.FrameLoop(ctx, gs, 32, 16, "compute")
    loop.frame()                                             # fine: megakernel
    with pytest.raises(hip.HipError, match="unknown kernel"):
        ctx.dispatch("nope", loop.renderpass.constants, 1, 1, 1)
    with pytest.raises(hip.HipError, match="no flattened layout"):
        ctx.dispatch("compute.wavefront", loop.renderpass.constants, 5, 3, 1)
    bad = loop.renderpass.constants.copy(); bad[8:16] = 0
    with pytest.raises(hip.HipError, match="bvh address"):
        ctx.dispatch("compute", bad, 5, 3, 1)
    bad = loop.renderpass.constants.copy(); bad[56:60] = np.frombuffer(np.uint32(4000).tobytes(), np.uint8)
    with pytest.raises(hip.HipError, match="output image"):
        ctx.dispatch("compute", bad, 5, 3, 1)
    gs.update_meta(sample_index=0)
    with pytest.raises(hip.HipError, match="sample_index"):
        ctx.dispatch("compute", loop.renderpass.constants, 5, 3, 1)
    gs.update_meta(sample_index=1, width=65536, height=65537)     # products of these index the queues: no 32-bit wrap-around
    with pytest.raises(hip.HipError, match="more than the path addresses"):
        ctx.dispatch("compute", loop.renderpass.constants, 5, 3, 1)
    gs.update_meta(width=32, height=16)
    with pytest.raises(hip.HipError, match="outside"):
        gs.scene_buffer.write(np.zeros(8, np.uint8), 316, 8)
    with pytest.raises(hip.HipError, match="LINEAR"):
        ctx.create_sampler(0, 0)
    with pytest.raises(hip.HipError, match="format"):
        ctx.create_2d_texture(4, 4, 44)
    ctx.dispatch("compute", loop.renderpass.constants, 0, 3, 1)   # empty dispatch is a no-op


def test_full_size_properties_1080p(ctx):
    """At benchmark size the oracle is too slow for a full comparison; check what must hold regardless:
    determinism (two runs identical), megakernel == wavefront except declared ties, oracle agreement on sampled rows."""
    b = make_bundle("street_exterior", 120000, 1920, 1080, max_bounce=3)
    wf1, gs, cap, _ = gpu_frames(ctx, b, 2, "compute.wavefront", capture=True)
    wf2, _, _, _ = gpu_frames(ctx, b, 2, "compute.wavefront", gpu_scene=gs)
    assert_same(wf1, wf2, "wavefront run-to-run")
    mega, _, capm, _ = gpu_frames(ctx, b, 2, "compute.megakernel", gpu_scene=gs, capture=True)
    differing = int((cap[0][..., 0] != capm[0][..., 0]).sum())
    assert differing <= 40, differing        # <= 2e-5 of 2.07 M primary rays
    # oracle on a few full rows (sample 2 needs sample 1's rows: render both samples on those rows only)
    import pyoracle
    o = pyoracle.OracleScene(dict(b, materials=gs.materials_host, textures=gs.texture_table))
    acc = np.zeros((1080, 1920, 4), np.float32); out = np.zeros_like(acc)
    rows = [(0, 2), (500, 502), (1078, 1080)]
    for s in (1, 2):
        o.set_meta(sample_index=s)
        for r in rows:
            o.render(acc, out, rows=r, use_flat=True)
        acc, out = out, acc
    for y0, y1 in rows:
        assert_same(wf1[y0:y1], acc[y0:y1], f"rows {y0}:{y1} vs oracle")


@pytest.mark.parametrize("kind,target,w,h,meta_kw", [("tiny", 0, 50, 37, dict(max_bounce=5)), ("atrium", 5000, 96, 54, dict(max_bounce=4)),
                                                     ("tiny", 0, 40, 24, dict(max_bounce=3, downscale_factor=2, enable_dof=1))])
def test_several_samples_per_dispatch_equal_as_many_single_sample_frames(ctx, kind, target, w, h, meta_kw):
    """"dispatch.samples" = S: one pass through the kernels for S samples, their accumulate chain in registers.  Seven
    samples as 3 + 3 + 1 and as 4 + 2 + 1 against the oracle's seven sequential frames, bit for bit; the counting variant
    sees the same rays, nodes and triangles as the oracle summed over the batch."""
    from path_tracer_amd import hip
    import pyoracle
    b = make_bundle(kind, target, w, h, **meta_kw)
    gs = hip.GpuScene(ctx, b, b["nodes"], b["flat"], b["meta"])
    ref, _, _ = oracle_for(b, gs, 7, use_flat=True)
    for per_dispatch in (3, 4):
        gs.set_meta(b["meta"])
        loop = hip.FrameLoop(ctx, gs, w, h, "compute", samples_per_dispatch=per_dispatch)
        loop.frames(per_dispatch)
        loop.frames(7 - per_dispatch)
        assert loop.sample_index == 8
        assert_same(loop.latest().read(), ref, f"{per_dispatch} samples per dispatch")
    # explicit dispatch sizes: 2 + 1 + 4, and sizes that do not add up are refused before anything is dispatched
    gs.set_meta(b["meta"])
    loop = hip.FrameLoop(ctx, gs, w, h, "compute")
    loop.frames(7, schedule=[2, 1, 4])
    assert loop.sample_index == 8
    assert_same(loop.latest().read(), ref, "dispatches of 2, 1 and 4 samples")
    with pytest.raises(ValueError):
        loop.frames(7, schedule=[3, 3])
    assert loop.sample_index == 8
    # work counters of one batch of three = the oracle's three frames added up
    gs.set_meta(b["meta"])
    loop = hip.FrameLoop(ctx, gs, w, h, "compute", samples_per_dispatch=3)
    ctx.stats_read(reset=True); ctx.stats_enable(True)
    loop.frames(3)
    st = ctx.stats_read(reset=True); ctx.stats_enable(False)
    o = pyoracle.OracleScene(dict(b, materials=gs.materials_host, textures=gs.texture_table))
    acc = np.zeros((h, w, 4), np.float32); out = np.zeros_like(acc)
    want = dict(rays=0, nodes=0, triangles=0, shaded=0)
    for s in (1, 2, 3):
        o.set_meta(sample_index=s)
        c = o.render(acc, out, use_flat=True)
        acc, out = out, acc
        for k in want:
            want[k] += c[k]
    assert {k: st[k] for k in want} == want and st["frames"] == 3


def test_several_samples_per_dispatch_with_the_partitions_and_where_it_is_refused(ctx):
    from path_tracer_amd import hip
    b = make_bundle("atrium", 4000, 80, 72, max_bounce=3)
    gs = hip.GpuScene(ctx, b, b["nodes"], b["flat"], b["meta"])
    whole, _, _ = oracle_for(b, gs, 4, use_flat=True)
    rows = (np.arange(72) // 8) % 2
    merged = np.zeros_like(whole)
    try:
        for rank in range(2):                                       # tile partition: each rank's stripes, four samples in one dispatch
            ctx.set_partition(rank, 2, 8)
            gs.set_meta(b["meta"])
            loop = hip.FrameLoop(ctx, gs, 80, 72, "compute", samples_per_dispatch=4)
            loop.frames(4)
            part = loop.latest().read()
            assert (part[rows != rank] == 0).all()
            merged[rows == rank] = part[rows == rank]
    finally:
        ctx.set_partition(0, 1, 8)
    assert_same(merged, whole, "tile partition, 4 samples per dispatch")
    try:                                                            # sample partition: rank 1 of 3 draws the global samples 2, 5, 8, 11
        ctx.set_sample_partition(1, 3)
        gs.set_meta(b["meta"])
        loop = hip.FrameLoop(ctx, gs, 80, 72, "compute", samples_per_dispatch=2)
        loop.frames(4)
        got = loop.latest().read()
    finally:
        ctx.set_sample_partition(0, 1)
    import pyoracle
    o = pyoracle.OracleScene(dict(b, materials=gs.materials_host, textures=gs.texture_table))
    acc = np.zeros((72, 80, 4), np.float32); out = np.zeros_like(acc)
    for k in range(1, 5):
        o.set_meta(sample_index=k)
        o.seed_sample_index = (k - 1) * 3 + 1 + 1
        o.render(acc, out, use_flat=True)
        acc, out = out, acc
    assert_same(got, acc, "sample partition, 2 samples per dispatch")
    # refused: the strict drop-in kernel, the heat-map, the per-pixel capture
    gs.set_meta(b["meta"])
    with pytest.raises(hip.HipError, match="dispatch.samples"):
        hip.FrameLoop(ctx, gs, 80, 72, "compute.megakernel").frame(2)
    gs.update_meta(debug_bvh=1)
    with pytest.raises(hip.HipError, match="dispatch.samples"):
        hip.FrameLoop(ctx, gs, 80, 72, "compute").frame(2)
    gs.update_meta(debug_bvh=0)
    cap = ctx.create_buffer(80 * 72 * 16)
    ctx.set_capture(cap)
    try:
        with pytest.raises(hip.HipError, match="capture"):
            hip.FrameLoop(ctx, gs, 80, 72, "compute").frame(2)
    finally:
        ctx.set_capture(None)
        cap.destroy()
    hip.FrameLoop(ctx, gs, 80, 72, "compute").frame(2)             # and fine again afterwards


def test_texture_extents_that_are_no_multiple_of_the_tile(ctx):
    """The shade step samples RGBA8 textures from a copy with one entry of 2 x 2 texels per lookup position, in 2 x 2 tiles; extents of 1, 3, 5, 10 ... texels leave partly
    filled tiles at the right and bottom edges, and REPEAT addressing wraps from them to the first column and row."""
    rng = np.random.default_rng(11)
    for (aw, ah), (mw, mh) in (((10, 6), (5, 3)), ((1, 1), (2, 7)), ((13, 4), (4, 13))):
        b = make_bundle("tiny", 0, 56, 36, max_bounce=4)
        b = dict(b, textures=[None, rng.integers(0, 256, (ah, aw, 4), dtype=np.uint8), rng.integers(0, 256, (mh, mw, 4), dtype=np.uint8)])
        for pipeline, use_flat in (("compute.wavefront", True), ("compute.megakernel", False)):
            img, gs, _, _ = gpu_frames(ctx, b, 2, pipeline)
            ref, _, _ = oracle_for(b, gs, 2, use_flat=use_flat)
            assert_same(img, ref, f"{pipeline}, textures {aw}x{ah} and {mw}x{mh}")
        again = rng.integers(0, 256, (ah, aw, 4), dtype=np.uint8)    # a second upload replaces the lookup copy too
        gs.textures[1].update(again)
        gs.texture_table[gs.textures[1].bindless_index] = again
        img, _, _, _ = gpu_frames(ctx, b, 2, "compute.wavefront", gpu_scene=gs)
        ref, _, _ = oracle_for(dict(b), gs, 2, use_flat=True)
        assert_same(img, ref, "after a second upload")


@pytest.mark.parametrize("width,height,world,stripe", [(40, 37, 3, 8), (17, 64, 4, 16), (8, 8, 2, 8), (33, 50, 8, 8)])
def test_stripe_pack_and_unpack_kernels(ctx, width, height, world, stripe):
    """ptr_image_pack_stripes / ptr_image_unpack_stripes against their definition: rank r's chunk holds the stripes r, r + world,
    ... in order, zero-padded to ceil(stripes / world) stripes; unpacking the chunks of all ranks rebuilds the frame."""
    from path_tracer_amd import hip
    rng = np.random.default_rng(5)
    frame = rng.random((height, width, 4), dtype=np.float32)
    src = ctx.create_2d_texture(width, height, hip.FORMAT_RGBA32_SFLOAT); src.update(frame)
    dst = ctx.create_2d_texture(width, height, hip.FORMAT_RGBA32_SFLOAT)
    stripes = (height + stripe - 1) // stripe
    chunk_stripes = (stripes + world - 1) // world
    chunk_floats = chunk_stripes * stripe * width * 4
    packed = ctx.create_buffer(chunk_floats * 4)
    gathered = ctx.create_buffer(chunk_floats * 4 * world)
    padded = np.zeros((chunk_stripes * world * stripe, width, 4), np.float32)
    padded[:height] = frame
    by_stripe = padded.reshape(chunk_stripes * world, stripe, width, 4)
    for r in range(world):
        ctx.pack_stripes(src, r, world, stripe, packed)
        got = packed.read(np.float32).reshape(chunk_stripes, stripe, width, 4)
        assert np.array_equal(got, by_stripe[r::world]), f"rank {r}"
        gathered.write(got, r * chunk_floats * 4)
    ctx.unpack_stripes(dst, world, stripe, gathered)
    assert np.array_equal(dst.read(), frame)
    small = ctx.create_buffer(16)
    with pytest.raises(hip.HipError, match="needed"):
        ctx.pack_stripes(src, 0, world, stripe, small)            # a buffer that cannot hold the chunk is refused
    small.destroy()
    for t in (src, dst):
        t.destroy()
    for b in (packed, gathered):
        b.destroy()


# ---- "dispatch.coalesce": the reference's frame loop as it is, rendered as batched dispatches ----
def _both_images(loop):
    return loop.latest().read(), loop.output.read()


@pytest.mark.parametrize("frames,hold", [(1, 8), (2, 8), (3, 8), (8, 8), (11, 8), (7, 2), (20, 64)])
def test_coalesced_dispatches_leave_both_images_as_single_dispatches_do(ctx, frames, hold):
    """K single-sample dispatches with the image swap after each, once as they are and once held back and rendered as dispatches
    of up to `hold` samples: the image written last AND the one written before it are equal bit for bit."""
    from path_tracer_amd import hip
    b = make_bundle("atrium", 30_000, 96, 72, max_bounce=3)
    gs = hip.GpuScene(ctx, b, b["nodes"], b["flat"], b["meta"])
    plain = hip.FrameLoop(ctx, gs, 96, 72, "compute")
    for _ in range(frames):
        plain.frame()
    want_last, want_other = _both_images(plain)
    held = hip.FrameLoop(ctx, gs, 96, 72, "compute")
    ctx.set_option("dispatch.coalesce", hold)
    ctx.profile_read(reset=True)
    ctx.profile_enable(True)
    try:
        for _ in range(frames):
            held.frame()
        got_last, got_other = _both_images(held)      # the read runs what is still held back
    finally:
        ctx.set_option("dispatch.coalesce", 1)
        ctx.profile_enable(False)
    # the frames really were rendered as ceil(frames / hold) dispatches
    assert ctx.profile_read(reset=True)["wavefront.generate"]["launches"] == -(-frames // hold)
    assert np.array_equal(got_last, want_last)
    assert np.array_equal(got_other, want_other)
    img, _, _ = oracle_frames(b, frames, use_flat=True, materials=gs.materials_host, textures=gs.texture_table)
    assert np.array_equal(got_last, img)


def test_coalescing_stops_where_the_chain_breaks(ctx):
    """A camera move, a read of the image in between, a write to a buffer the frames read, a restart of the accumulation, a
    downscaled frame: each ends the batch being held back, and the images are those of the plain loop."""
    from path_tracer_amd import hip
    b = make_bundle("street_interior", 30_000, 80, 64, max_bounce=3)
    gs = hip.GpuScene(ctx, b, b["nodes"], b["flat"], b["meta"])

    def run(hold):
        loop = hip.FrameLoop(ctx, gs, 80, 64, "compute")
        gs.set_meta(b["meta"])
        ctx.set_option("dispatch.coalesce", hold)
        seen = []
        try:
            for _ in range(3):
                loop.frame()
            gs.update_meta(max_bounce=2)                       # another scene state: not the same chain any more
            for _ in range(3):
                loop.frame()
            seen.append(loop.latest().read())                  # a read in the middle of a batch
            for _ in range(2):
                loop.frame()
            mats = gs.materials_buffer.read(np.uint8)
            gs.materials_buffer.write(mats)                    # a write to a buffer the held-back frames read
            for _ in range(2):
                loop.frame()
            loop.reset()                                       # sample_index back to 1: a new chain on the old images
            for _ in range(4):
                loop.frame()
            gs.update_meta(downscale_factor=2)
            for _ in range(3):
                loop.frame()
            seen.extend(_both_images(loop))
        finally:
            ctx.set_option("dispatch.coalesce", 1)
            gs.set_meta(b["meta"])
        return seen

    for want, got in zip(run(1), run(6)):
        assert np.array_equal(want, got)


def test_a_held_back_chain_stays_within_what_a_batched_dispatch_can_render(ctx):
    """Images of unequal extent are no chain at all (a batched dispatch needs equal extents, single dispatches do not); a scratch
    budget shorter than the chain cuts it into several batched dispatches; a batched dispatch that is refused is replayed one
    dispatch at a time.  The images are those of the plain loop every time."""
    from path_tracer_amd import hip
    b = make_bundle("atrium", 20_000, 64, 48, max_bounce=3)
    gs = hip.GpuScene(ctx, b, b["nodes"], b["flat"], b["meta"])
    want, _, _ = oracle_frames(b, 6, use_flat=True, materials=gs.materials_host, textures=gs.texture_table)

    def generate_launches(body):
        ctx.profile_read(reset=True)
        ctx.profile_enable(True)
        try:
            body()
        finally:
            ctx.profile_enable(False)
        return ctx.profile_read(reset=True)["wavefront.generate"]["launches"]

    # (1) a taller accumulation image: every dispatch runs on its own
    tall = (ctx.create_2d_texture(64, 48, hip.FORMAT_RGBA32_SFLOAT), ctx.create_2d_texture(64, 56, hip.FORMAT_RGBA32_SFLOAT))
    loop = hip.FrameLoop(ctx, gs, 64, 48, "compute", textures=tall)
    ctx.set_option("dispatch.coalesce", 8)
    try:
        n = generate_launches(lambda: [loop.frame() for _ in range(6)] and ctx.flush())
        assert n == 6
        assert np.array_equal(loop.latest().read()[:48], want)
        # (2) a budget of 1 MB holds 64 x 48 x 132 B twice: chains of two
        ctx.set_option("dispatch.coalesce_scratch_mb", 1)
        loop = hip.FrameLoop(ctx, gs, 64, 48, "compute")
        n = generate_launches(lambda: [loop.frame() for _ in range(6)] and ctx.flush())
        assert n == 3
        assert np.array_equal(loop.latest().read(), want)
        ctx.set_option("dispatch.coalesce_scratch_mb", 8192)
        # (3) the batched dispatch fails: the six frames still arrive, one dispatch each
        ctx.set_option("debug.refuse_batches", 1)
        loop = hip.FrameLoop(ctx, gs, 64, 48, "compute")
        n = generate_launches(lambda: [loop.frame() for _ in range(6)] and ctx.flush())
        assert n == 6
        last, other = _both_images(loop)
        assert np.array_equal(last, want)
        want5, _, _ = oracle_frames(b, 5, use_flat=True, materials=gs.materials_host, textures=gs.texture_table)
        assert np.array_equal(other, want5)
    finally:
        ctx.set_option("debug.refuse_batches", 0)
        ctx.set_option("dispatch.coalesce_scratch_mb", 8192)
        ctx.set_option("dispatch.coalesce", 1)
    for t in tall:
        t.destroy()


def test_a_dispatch_that_cannot_run_is_refused_at_once_also_when_dispatches_are_held_back(ctx):
    from path_tracer_amd import hip
    b = make_bundle("tiny", 0, 32, 24, max_bounce=2)
    gs = hip.GpuScene(ctx, b, b["nodes"], None, b["meta"])               # no flattened layout bound
    loop = hip.FrameLoop(ctx, gs, 32, 24, "compute.wavefront")
    ctx.set_option("dispatch.coalesce", 8)
    try:
        with pytest.raises(hip.HipError, match="no flattened layout"):
            loop.frame()                                                  # reported by the dispatch itself, not by a later call
        ctx.sync()                                                        # nothing was left behind
    finally:
        ctx.set_option("dispatch.coalesce", 1)


def test_a_held_back_dispatch_runs_before_anything_else_sees_its_images(ctx):
    from path_tracer_amd import hip
    b = make_bundle("atrium", 20_000, 64, 48, max_bounce=2)
    gs = hip.GpuScene(ctx, b, b["nodes"], b["flat"], b["meta"])
    loop = hip.FrameLoop(ctx, gs, 64, 48, "compute")
    tone = ctx.create_2d_texture(64, 48, hip.FORMAT_RGBA32_SFLOAT)
    rp = ctx.create_compute_renderpass()
    rp.set_pipeline("tonemapping")
    rp.set_constant(0, gs.scene_buffer)
    rp.set_ouput_texture(tone)
    rp.set_dispatch_size((64 + 7) // 8, (48 + 7) // 8, 1)
    ctx.set_option("dispatch.coalesce", 16)
    try:
        for _ in range(5):
            loop.frame()
        rp.set_constant(60, loop.latest())
        rp.execute()                                           # another kernel: the five frames it reads run first
        got5 = tone.read()
        for _ in range(3):
            loop.frame()
        ctx.flush()                                            # enqueued now, not waited for
        rp.set_constant(60, loop.latest())
        rp.execute()
        got8 = tone.read()
    finally:
        ctx.set_option("dispatch.coalesce", 1)
    for frames, got in ((5, got5), (8, got8)):
        want, _, _ = oracle_frames(b, frames, use_flat=True, materials=gs.materials_host, textures=gs.texture_table)
        assert np.array_equal(got[..., :3], np.sqrt(want[..., :3])), frames
    tone.destroy()


# ---- "bvh.auto_flatten": the reference's buffers in, the fast pipeline out ----
def test_dispatch_flattens_the_tree_itself_when_asked_to(ctx):
    """A strict drop-in caller uploads the reference's seven buffers and binds nothing else.  With "bvh.auto_flatten" its
    "compute" dispatches build the flattened layout on the GPU (tie ranks from the threaded tree), bind it and take the wavefront
    pipeline: same image as the megakernel over the reference's tree, which is what the dispatch runs without the option."""
    from path_tracer_amd import hip
    b = make_bundle("atrium", 20_000, 96, 64, max_bounce=3)
    gs = hip.GpuScene(ctx, b, b["nodes"], None, b["meta"])              # no flattened layout uploaded or bound

    def render(spp):
        loop = hip.FrameLoop(ctx, gs, 96, 64, "compute")
        ctx.profile_read(reset=True)
        ctx.profile_enable(True)
        for _ in range(spp):
            loop.frame()
        img = loop.latest().read()
        ctx.profile_enable(False)
        ran = {k for k, v in ctx.profile_read(reset=True).items() if v["launches"]}
        loop.accumulation.destroy(); loop.output.destroy()
        return img, ran

    plain, ran = render(3)
    assert "compute.megakernel" in ran and "wavefront.extend" not in ran
    ctx.set_option("bvh.auto_flatten", 1)
    try:
        fast, ran = render(3)
        assert "wavefront.extend" in ran and "bvh.build_flat" in ran and "compute.megakernel" not in ran
        assert np.array_equal(fast, plain)
        again, ran = render(3)                                            # built once: the layout stays bound
        assert "bvh.build_flat" not in ran and np.array_equal(again, plain)
        ctx.set_option("dispatch.coalesce", 2)                            # with held-back dispatches on top: 3 = 2 + 1
        both, ran = render(3)
        ctx.set_option("dispatch.coalesce", 1)
        assert "wavefront.extend" in ran and np.array_equal(both, plain)

        # new geometry in the same buffers (the scene mirrored in x: other tree, other image): the layout is rebuilt from it
        m = dict(b)
        m["positions"] = np.asarray(b["positions"], np.float32).reshape(-1, 3) * np.array([-1, 1, 1], np.float32)
        m["triangles"] = np.asarray(b["triangles"], np.float32).reshape(-1, 3, 3) * np.array([-1, 1, 1], np.float32)
        from path_tracer_amd import host
        nodes = host.bvh_build(m["triangles"].reshape(-1, 9))
        assert nodes.nbytes == np.asarray(b["nodes"]).nbytes
        gs.positions_buffer.write(np.ascontiguousarray(m["positions"], np.float32))
        gs.bvh_buffer.write(nodes)
        moved, ran = render(2)
        assert "bvh.build_flat" in ran and "wavefront.extend" in ran
    finally:
        ctx.set_option("bvh.auto_flatten", 0)
    ctx.bind_flat(gs.bvh_buffer, None)                                    # back to the reference's tree alone
    want, ran = render(2)
    assert "compute.megakernel" in ran and np.array_equal(moved, want)
    assert not np.array_equal(moved, plain[:])


def test_present_style_readback_converts_on_the_device(ctx):
    """ptr_image_read_unorm8: float -> 8-bit UNORM as the reference's present blit converts (NaN 0, clamp, x 255, round to nearest)."""
    from path_tracer_amd import hip
    rng = np.random.default_rng(11)
    W, H = 77, 45
    img = rng.uniform(-0.25, 1.25, size=(H, W, 4)).astype(np.float32)
    img[0, 0] = [np.nan, np.inf, -np.inf, 0.5]
    img[0, 1] = [0.0, 1.0, 0.5 / 255.0, 1.5 / 255.0]               # the two ties round to even: 0 and 2
    img[0, 2] = np.float32(254.5 / 255.0)
    t = ctx.create_2d_texture(W, H, hip.FORMAT_RGBA32_SFLOAT)
    t.update(img)
    got = t.read_unorm8()
    clean = np.where(np.isnan(img), np.float32(0), img)
    want = np.rint(np.clip(clean, np.float32(0), np.float32(1)) * np.float32(255.0)).astype(np.uint8)
    assert np.array_equal(got, want)
    assert got[0, 0].tolist() == [0, 255, 0, 128] and got[0, 1].tolist()[:2] == [0, 255]
    small = ctx.create_2d_texture(8, 8, hip.FORMAT_RGBA8_UNORM)
    with pytest.raises(hip.HipError, match="RGBA32_SFLOAT"):
        small.read_unorm8()
    for x in (t, small):
        x.destroy()


@pytest.mark.parametrize("turns", [0, 1000000])
@pytest.mark.parametrize("bounces", [1, 2, 3, 5])
def test_last_bounce_stops_at_the_first_hit_and_the_image_does_not_change(ctx, bounces, turns):
    """The traversal of a path's last bounce ends at the first hit it finds and its shade step only asks whether the ray escaped
    (compute.comp:130-136, 241-244: a path that hit something on its last bounce contributes nothing, whatever it hit) -- unless
    the per-pixel capture is on, which wants the closest hit.  Images with and without the capture equal the oracle's."""
    from test_spheres import sphere_set
    from common import add_spheres
    for kind, target, w, h in (("tiny", 0, 50, 37), ("atrium", 5000, 96, 54)):
        b = make_bundle(kind, target, w, h, max_bounce=bounces, min_bounce=0)
        if kind == "tiny":
            b = add_spheres(b, sphere_set())
        ctx.set_option("extend.alone_turns", turns)
        try:
            img, gs, _, _ = gpu_frames(ctx, b, 3, "compute.wavefront")
            img_cap, _, cap, _ = gpu_frames(ctx, b, 3, "compute.wavefront", gpu_scene=gs, capture=True)
        finally:
            ctx.set_option("extend.alone_turns", 0)
        ref, _, aov = oracle_frames(b, 3, use_flat=True, aov=True, materials=gs.materials_host, textures=gs.texture_table,
                                    spheres=gs.spheres_host)
        assert_same(img, ref, f"{kind}, {bounces} bounces, first hit on the last one")
        assert_same(img_cap, ref, f"{kind}, {bounces} bounces, with the capture (closest hits throughout)")
        assert_same(cap[0], aov[0], "captured primitive ids")
