"""The measurement configurations of the hot-path contract at their real sizes, as far as one GPU and a CPU oracle
that traces ~0.5 Mrays/s allow:
  C2  atrium (Sponza-class) 1920x1080, max_bounce 2: primitive id and t of the first two hits against the oracle's own
      traversal order on every pixel (ties counted, not hidden), and the image bit for bit;
  C3  street interior 1920x1080, max_bounce 4, several samples: the oracle on sampled row bands, bit for bit
      (stronger than the RMSE bound the contract asks for);
  headline  the benchmark's own scene at the benchmark's own size, against the oracle on row bands, bit for bit;
  C1  atrium 1280x720, max_bounce 4: both pipelines against the oracle on row bands;
  C4  street exterior (2.8 M triangles) 3840x2160, 64 samples, tile mode over 8 ranks: the ranks' stripes, packed and
      unpacked by the library's stripe kernels, rebuild the single-GPU frame bit for bit (the ranks are eight partitions of
      one context here; the collective runs in the multi-process tests);
  C5  100.7 M triangles as placed copies, one rank of eight of the sample partition against the oracle."""
import numpy as np
import pytest

from common import gpu_frames, make_bundle, oracle_frames

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from path_tracer_amd import hip
    c = hip.Context(0)
    yield c
    c.close()


def oracle_bands(bundle, gs, bands, spp, use_flat=True):
    """The oracle on a few row bands of a frame too large to render whole on the CPU: `spp` samples, each band's accumulation
    chain kept (sample s needs the rows sample s-1 left behind)."""
    import pyoracle
    o = pyoracle.OracleScene(dict(bundle, materials=gs.materials_host, textures=gs.texture_table))
    H, W = bundle["height"], bundle["width"]
    acc = np.zeros((H, W, 4), np.float32)
    out = np.zeros_like(acc)
    for s in range(1, spp + 1):
        o.set_meta(sample_index=s)
        for r in bands:
            o.render(acc, out, rows=r, use_flat=use_flat, threads=16)
        acc, out = out, acc
    return acc


def assert_bands_equal(img, ref, bands, what):
    for y0, y1 in bands:
        bad = int((img[y0:y1] != ref[y0:y1]).any(axis=-1).sum())
        assert bad == 0, f"{what}: {bad} pixels of rows {y0}:{y1} differ from the oracle"


def test_headline_workload_street_exterior_2_8m_triangles_1080p(ctx):
    """The exact workload bench.py times (its own scene builder with its default arguments: 2 796 540 triangles, 512^2
    textures, 1920x1080, max_bounce 4, "compute" with two frames in flight), three samples, against the oracle on six row
    bands spread over the frame -- bit for bit."""
    import os
    import sys
    from path_tracer_amd import hip
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    args = bench.make_parser().parse_args([])
    sc, nodes, flat, meta, _ = bench.build_scene(args, rank=1)
    assert sc["triangles"].shape[0] == 2_796_540 and (args.width, args.height, args.bounces) == (1920, 1080, 4)
    b = dict(sc, nodes=nodes, flat=flat, meta=meta, width=args.width, height=args.height)
    gs = hip.GpuScene(ctx, b, nodes, flat, meta)
    spp = 3
    img, _, _, st = gpu_frames(ctx, b, spp, args.pipeline, gpu_scene=gs)
    bands = [(0, 8), (200, 208), (416, 424), (640, 648), (856, 864), (1072, 1080)]
    assert_bands_equal(img, oracle_bands(b, gs, bands, spp), bands, "headline workload")
    assert np.isfinite(img).all() and (img[..., 3] == 1.0).all()
    # the counting variant sees the same work the benchmark line reports: about 5.15 M rays per frame
    _, _, _, st = gpu_frames(ctx, b, 1, args.pipeline, gpu_scene=gs, stats=True)
    assert 5.0e6 < st["rays"] < 5.3e6, st


def test_c1_atrium_720p_four_bounces(ctx):
    """C1's workload (Sponza-class atrium, 262 k triangles, 1280x720, 1 spp, max_bounce 4) on the HIP path: the strict drop-in
    kernel against the oracle's walk of the reference's own tree, the fast path against the oracle's walk of the flattened
    layout, both on row bands, bit for bit."""
    b = make_bundle("atrium", 262144, 1280, 720, max_bounce=4)
    assert 250_000 < b["triangles"].shape[0] < 275_000
    bands = [(0, 4), (180, 184), (356, 364), (540, 544), (716, 720)]
    img, gs, _, _ = gpu_frames(ctx, b, 1, "compute")
    assert_bands_equal(img, oracle_bands(b, gs, bands, 1, use_flat=True), bands, "C1 wavefront")
    mega, _, _, _ = gpu_frames(ctx, b, 1, "compute.megakernel", gpu_scene=gs)
    thin = [(0, 2), (358, 360), (718, 720)]                        # the reference-order tree costs ~670 nodes per ray on the CPU
    assert_bands_equal(mega, oracle_bands(b, gs, thin, 1, use_flat=False), thin, "C1 megakernel")


def test_c2_atrium_1080p_first_two_hits_and_image(ctx):
    b = make_bundle("atrium", 262144, 1920, 1080, max_bounce=2)
    assert 250_000 < b["triangles"].shape[0] < 275_000
    img, gs, cap, st = gpu_frames(ctx, b, 1, "compute.wavefront", capture=True, stats=True)
    mega, _, capm, _ = gpu_frames(ctx, b, 1, "compute.megakernel", gpu_scene=gs, capture=True)
    # the oracle walks the reference's threaded tree in the reference's order (what the megakernel does too)
    ref, counters, aov = oracle_frames(b, 1, materials=gs.materials_host, textures=gs.texture_table, aov=True, threads=16)
    assert np.array_equal(mega, ref) and np.array_equal(capm[0], aov[0]) and np.array_equal(capm[1], aov[1])
    # the fast path walks another tree: equal ids except the declared tie classes, equal t wherever the ids agree
    for bounce in (0, 1):
        same = cap[0][..., bounce] == aov[0][..., bounce]
        assert int((~same).sum()) <= 40, int((~same).sum())
        assert np.array_equal(cap[1][..., bounce][same], aov[1][..., bounce][same])
    assert counters["nan_hits"] == 0
    differing = np.abs(img - ref).max(axis=2) > 0
    assert int(differing.sum()) <= 80, int(differing.sum())      # only pixels whose path went through a tie


def test_c3_street_interior_1080p_sampled_bands(ctx):
    spp = 16                                                       # the sample count the configuration states
    b = make_bundle("street_interior", 1_000_000, 1920, 1080, max_bounce=4)
    assert 950_000 < b["triangles"].shape[0] < 1_050_000
    img, gs, _, _ = gpu_frames(ctx, b, spp, "compute.wavefront")
    bands = [(0, 4), (268, 272), (540, 544), (812, 816), (1076, 1080)]
    assert_bands_equal(img, oracle_bands(b, gs, bands, spp), bands, "C3, 16 samples")
    assert np.isfinite(img).all() and img[..., :3].min() >= 0.0 and img[..., :3].max() <= 1.0 and (img[..., 3] == 1.0).all()


def test_c4_street_exterior_4k_64spp_tile_mode_over_eight_ranks(ctx):
    """C4 at its size: the benchmark's 2.8 M-triangle street, 3840x2160, 64 samples (dispatches of 16), tiles over 8 ranks.
    One GPU plays the ranks one after the other; every rank's stripes are packed by ptr_image_pack_stripes, the chunks are
    laid side by side the way an all-gather delivers them, and ptr_image_unpack_stripes rebuilds the frame: bit-identical to
    the frame one GPU renders alone, which in turn equals the oracle on row bands.  (The collective itself runs in the
    multi-process tests; on the wire it moves exactly these chunks.)"""
    import os
    import sys
    from path_tracer_amd import hip
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    W, H, world, spp, per_dispatch, stripe = 3840, 2160, 8, 64, 16, 8
    args = bench.make_parser().parse_args(["--width", str(W), "--height", str(H)])
    sc, nodes, flat, meta, _ = bench.build_scene(args, rank=1)
    assert sc["triangles"].shape[0] == 2_796_540
    b = dict(sc, nodes=nodes, flat=flat, meta=meta, width=W, height=H)
    gs = hip.GpuScene(ctx, b, nodes, flat, meta)
    loop = hip.FrameLoop(ctx, gs, W, H, "compute", samples_per_dispatch=per_dispatch)

    def render():
        loop.reset()
        for t in (loop.accumulation, loop.output):
            t.update(np.zeros((H, W, 4), np.float32))
        loop.frames(spp)
        return loop.latest()

    whole = render().read()
    bands = [(0, 2), (1079, 1081), (2158, 2160)]
    assert_bands_equal(whole, oracle_bands(b, gs, bands, spp), bands, "C4, one GPU, 64 samples")

    chunk_bytes = -(-((H + stripe - 1) // stripe) // world) * stripe * W * 16
    packed = ctx.create_buffer(chunk_bytes)
    gathered = ctx.create_buffer(chunk_bytes * world)
    merged = ctx.create_2d_texture(W, H, hip.FORMAT_RGBA32_SFLOAT)
    stripe_of_row = (np.arange(H) // stripe) % world
    try:
        for rank in range(world):
            ctx.set_partition(rank, world, stripe)
            latest = render()
            if rank == 3:
                part = latest.read()
                assert (part[stripe_of_row != rank] == 0).all()   # a rank never writes foreign rows
            ctx.pack_stripes(latest, rank, world, stripe, packed)
            gathered.write(packed.read(np.uint8), rank * chunk_bytes)
    finally:
        ctx.set_partition(0, 1, stripe)
    ctx.unpack_stripes(merged, world, stripe, gathered)
    assert np.array_equal(merged.read(), whole)
    for t in (merged, loop.accumulation, loop.output):
        t.destroy()
    for buf in (packed, gathered):
        buf.destroy()


def test_c5_100m_triangles_as_36_placed_copies_one_rank_of_eight(ctx):
    """C5 at its size: the benchmark's scene placed 6 x 6 times through a top-level tree (100.7 M triangles), 1920x1080,
    max_bounce 4, the samples dealt to 8 ranks.  This GPU plays rank 5: its frames -- global samples 6 and 14 -- against the
    oracle's two-level traversal seeded with the same sample numbers, on row bands, bit for bit.  (What the ranks then do with
    their images -- linearise, all-reduce, resolve -- is pinned at small size in the parity tests and on CPU ranks.)"""
    import os
    import sys
    import pyoracle
    from path_tracer_amd import hip
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    args = bench.make_parser().parse_args(["--instances", "6x6", "--two-level"])
    sc, nodes, flat, meta, _ = bench.build_scene(args, rank=1)
    assert sc["triangles"].shape[0] * 36 == 100_675_440 and sc["flat_instances"] is not None
    b = dict(sc, nodes=nodes, flat=flat, meta=meta, width=args.width, height=args.height)
    gs = hip.GpuScene(ctx, b, nodes, flat, meta)
    world, rank, frames = 8, 5, 2
    H, W = args.height, args.width
    try:
        ctx.set_sample_partition(rank, world)
        img, _, _, _ = gpu_frames(ctx, b, frames, "compute", gpu_scene=gs)
    finally:
        ctx.set_sample_partition(0, 1)
    o = pyoracle.OracleScene(dict(b, materials=gs.materials_host, textures=gs.texture_table))
    acc = np.zeros((H, W, 4), np.float32)
    out = np.zeros_like(acc)
    bands = [(0, 4), (300, 304), (538, 542), (800, 804), (1076, 1080)]
    hit_copies = set()
    for k in range(1, frames + 1):
        o.set_meta(sample_index=k)
        o.seed_sample_index = (k - 1) * world + rank + 1
        for r in bands:
            cap = (np.full((H, W, 2), -1, np.int32), np.zeros((H, W, 2), np.float32))
            o.render(acc, out, rows=r, use_flat=True, threads=16, aov=cap)
            ids = cap[0][r[0]:r[1], :, 0]
            hit_copies |= set(np.unique(ids[ids >= 0] // sc["triangles"].shape[0]).tolist())
        acc, out = out, acc
    assert len(hit_copies) >= 6, hit_copies                          # the bands see many of the copies
    assert_bands_equal(img, acc, bands, "C5, rank 5 of 8")


def test_c5_class_25m_triangles_written_out_multi_gb_tree(ctx):
    """The multi-GB-tree case of C5 with the tree written out instead of placed: 3 x 3 copies of the benchmark's scene as ONE
    flat scene in the 4-word index layout (25.2 M triangles: 1.6 GB of triangle records + 0.4 GB of nodes resident in HBM, far
    past L2 + Infinity Cache), 1920x1080, max_bounce 4, two samples, against the oracle walking the same layout on six row
    bands, bit for bit."""
    import os
    import sys
    from path_tracer_amd import hip
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    args = bench.make_parser().parse_args(["--instances", "3x3"])
    sc, nodes, flat, meta, _ = bench.build_scene(args, rank=1)
    assert sc["triangles"].shape[0] == 9 * 2_796_540 and flat.nbytes > 1_900_000_000
    b = dict(sc, nodes=nodes, flat=flat, meta=meta, width=args.width, height=args.height)
    gs = hip.GpuScene(ctx, b, nodes, flat, meta)
    frames = 2
    img, _, _, _ = gpu_frames(ctx, b, frames, "compute", gpu_scene=gs)
    bands = [(0, 3), (215, 218), (431, 434), (647, 650), (863, 866), (1077, 1080)]
    want = oracle_bands(b, gs, bands, frames)
    assert_bands_equal(img, want, bands, "25.2 M triangles written out")
    assert float(img[..., :3].std()) > 0.05                              # a picture, not a constant


def test_c5_placed_copies_sample_partition_vs_sequential(ctx):
    """C5's shape at a size the CPU can follow: a scene of placed copies, the samples dealt out to 4 ranks (four partitions of
    one context here), every rank's running mean linearised, summed (what the all-reduce does) and resolved.  The result is
    not bit-equal to 16 sequential samples on one GPU -- each frame's clamp + sRGB round trip is nonlinear -- so it is held to
    an RMSE bound on linear luminance, and the ranks' partial images are checked to be different samples of the same picture."""
    from path_tracer_amd import hip, host, scenes
    world, per_rank = 4, 4
    W, H = 320, 180
    base = make_bundle("street_exterior", 60000, W, H, max_bounce=4)
    m = scenes.grid_transforms(base, 3, 3)
    two = dict(base, flat_instances=host.bvh_flatten_instances(base["flat"], m))
    gs = hip.GpuScene(ctx, two, two["nodes"], two["flat"], two["meta"])

    def srgb_to_linear(c):
        c = np.clip(c.astype(np.float64), 0, 1)
        return np.where(c < 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)

    def luminance(rgb):
        return rgb @ np.array([0.2126, 0.7152, 0.0722])

    linear = []
    try:
        for rank in range(world):
            ctx.set_sample_partition(rank, world)
            gs.set_meta(two["meta"])
            loop = hip.FrameLoop(ctx, gs, W, H, "compute.wavefront")
            for _ in range(per_rank):
                loop.frame()
            rp = ctx.create_compute_renderpass()
            rp.set_pipeline("linearize")
            rp.set_ouput_texture(loop.output)
            rp.set_constant(60, loop.latest())
            rp.set_dispatch_size(1, 1, 1)
            rp.execute()
            linear.append(loop.output.read())
        total = np.sum(linear, axis=0).astype(np.float32)
        assert (total[..., 3] == world).all()
        loop.output.update(total)
        rp.set_pipeline("resolve")
        rp.set_ouput_texture(loop.latest())
        rp.set_constant(60, loop.output)
        rp.execute()
        combined = loop.latest().read()
    finally:
        ctx.set_sample_partition(0, 1)
    gs.set_meta(two["meta"])
    seq_loop = hip.FrameLoop(ctx, gs, W, H, "compute.wavefront")
    for _ in range(world * per_rank):
        seq_loop.frame()
    sequential = seq_loop.latest().read()
    la, lb = luminance(srgb_to_linear(combined[..., :3])), luminance(srgb_to_linear(sequential[..., :3]))
    rmse = float(np.sqrt(np.mean((la - lb) ** 2)))
    assert rmse < 0.02, rmse
    assert abs(la.mean() - lb.mean()) < 0.004, (la.mean(), lb.mean())
    # the partial images are different samples (not four copies of one), each noisier than the combination
    l0, l1 = luminance(linear[0][..., :3].astype(np.float64)), luminance(linear[1][..., :3].astype(np.float64))
    assert float(np.sqrt(np.mean((l0 - l1) ** 2))) > 2 * rmse
