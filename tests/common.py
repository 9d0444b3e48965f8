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


def oracle_frames(bundle, spp, use_flat=False, materials=None, textures=None, aov=False, threads=None, first_sample=1):
    """Renders sample_index = first_sample .. first_sample+spp-1 with the CPU oracle, ping-ponging the two images
    like src/main.cpp:406-416.  Returns (image, counters of the last frame, (aov_prim, aov_t) of the last frame)."""
    import pyoracle
    arrays = dict(bundle)
    if materials is not None:
        arrays["materials"] = materials
    if textures is not None:
        arrays["textures"] = textures
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
