"""ptr_dispatch_static -- the "compute" dispatch in the library's fixed per-GPU scratch array, for callers that cannot
allocate -- against ptr_dispatch on the same inputs, through the C ABI.  Images are compared for equality (numpy `==`),
like every other parity test of the path: both entry points run the same kernels on the same fp32 arithmetic."""
import numpy as np
import pytest

from common import make_bundle, oracle_frames

pytestmark = pytest.mark.gpu


def render(ctx, gs, width, height, spp, static, pipeline="compute", per_dispatch=1):
    """`spp` frames of the reference's frame protocol (src/main.cpp:406-416) through one of the two entry points."""
    from path_tracer_amd import hip
    loop = hip.FrameLoop(ctx, gs, width, height, pipeline)
    if static:
        loop.renderpass.execute = lambda: ctx.dispatch_static(pipeline, loop.renderpass.constants, *loop.renderpass.group_count)
    loop.samples_per_dispatch = per_dispatch
    loop.frames(spp)
    return loop.latest().read()


def test_static_dispatch_equals_the_allocating_dispatch_twice_in_a_row_the_second_smaller():
    from path_tracer_amd import hip
    ctx = hip.Context(0)
    try:
        big = make_bundle("atrium", 6000, 200, 120, max_bounce=4)
        small = make_bundle("tiny", 0, 56, 40, max_bounce=3)          # other scene, fewer pixels: shorter queues, stale tail behind them
        gs_big = hip.GpuScene(ctx, big, big["nodes"], big["flat"], big["meta"])
        gs_small = hip.GpuScene(ctx, small, small["nodes"], small["flat"], small["meta"])

        # nothing has looked at these buffers yet, and looking allocates: the static entry point says so instead of guessing
        with pytest.raises(hip.HipError, match="have not been checked|has not been read yet"):
            render(ctx, gs_big, 200, 120, 1, static=True)

        want_big = render(ctx, gs_big, 200, 120, 3, static=False)     # also checks the scene and gathers the shading records
        want_small = render(ctx, gs_small, 56, 40, 3, static=False)
        ctx.sync()

        got_big = render(ctx, gs_big, 200, 120, 3, static=True)       # first call ... (three frames each: the accumulation chain)
        got_small = render(ctx, gs_small, 56, 40, 3, static=True)     # ... and a second, smaller one right behind it
        assert np.array_equal(got_big, want_big), int((got_big != want_big).any(axis=-1).sum())
        assert np.array_equal(got_small, want_small), int((got_small != want_small).any(axis=-1).sum())

        # and both equal the oracle, so "equal to each other" is not two wrong pictures agreeing
        ref, _, _ = oracle_frames(small, 3, use_flat=True, materials=gs_small.materials_host, textures=gs_small.texture_table)
        assert np.array_equal(got_small, ref)

        # the strict drop-in kernel needs no scratch at all; the static entry point launches it too
        mega = render(ctx, gs_small, 56, 40, 2, static=False, pipeline="compute.megakernel")
        assert np.array_equal(render(ctx, gs_small, 56, 40, 2, static=True, pipeline="compute.megakernel"), mega)

        # several samples per dispatch ("dispatch.samples") through the fixed scratch: 3 x 2 samples of the small scene
        assert np.array_equal(render(ctx, gs_small, 56, 40, 6, static=True, per_dispatch=2), render(ctx, gs_small, 56, 40, 6, static=False))

        # back and forth between the two entry points on one context: neither disturbs the other
        assert np.array_equal(render(ctx, gs_big, 200, 120, 2, static=False), render(ctx, gs_big, 200, 120, 2, static=True))

        # dispatches held back under "dispatch.coalesce" would have to allocate when they run: the static entry point does not
        # run them on the quiet, it says what to call first
        ctx.set_option("dispatch.coalesce", 8)
        held = hip.FrameLoop(ctx, gs_small, 56, 40, "compute")
        held.frame()
        with pytest.raises(hip.HipError, match="held back"):
            ctx.dispatch_static("compute", held.renderpass.constants, *held.renderpass.group_count)
        ctx.flush()
        ctx.dispatch_static("compute", held.renderpass.constants, *held.renderpass.group_count)
        ctx.set_option("dispatch.coalesce", 1)
    finally:
        ctx.close()


def test_static_dispatch_refuses_what_its_fixed_scratch_cannot_hold():
    from path_tracer_amd import hip
    ctx = hip.Context(0)
    try:
        b = make_bundle("tiny", 0, 640, 360, max_bounce=2)            # 230 400 pixels: more than 163 840 queue slots
        gs = hip.GpuScene(ctx, b, b["nodes"], b["flat"], b["meta"])
        render(ctx, gs, 640, 360, 1, static=False)
        with pytest.raises(hip.HipError, match="queue slots"):
            render(ctx, gs, 640, 360, 1, static=True)
        small = dict(b, meta=make_bundle("tiny", 0, 64, 40, max_bounce=2)["meta"], width=64, height=40)
        gs.set_meta(small["meta"])                                     # the same scene at a size that fits goes through
        assert np.array_equal(render(ctx, gs, 64, 40, 2, static=True), render(ctx, gs, 64, 40, 2, static=False))
    finally:
        ctx.close()


def free_device_memory():
    """Free bytes of the current device as the HIP runtime reports them."""
    import ctypes
    runtime = ctypes.CDLL("libamdhip64.so")
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert runtime.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value


def test_static_dispatch_allocates_nothing_and_does_not_wait():
    """The property the entry point exists for: free device memory is the same before and after a run of static dispatches
    (larger frames and another scene than the allocating entry point has seen on this context included), and what would need a
    copy back to the host is refused, not done."""
    from path_tracer_amd import hip
    ctx = hip.Context(0)
    try:
        a = make_bundle("atrium", 6000, 96, 64, max_bounce=4)
        b = make_bundle("street_interior", 8000, 320, 200, max_bounce=3)
        gs_a = hip.GpuScene(ctx, a, a["nodes"], a["flat"], a["meta"])
        gs_b = hip.GpuScene(ctx, b, b["nodes"], b["flat"], b["meta"])
        # the allocating entry point checks both scenes, reads their headers, gathers the records -- on SMALL frames of scene b
        render(ctx, gs_a, 96, 64, 1, static=False)
        small_b = make_bundle("street_interior", 8000, 64, 40, max_bounce=3)["meta"]
        gs_b.set_meta(small_b)
        render(ctx, gs_b, 64, 40, 1, static=False)
        gs_b.set_meta(b["meta"])
        loops = []
        for gs, (w, h) in ((gs_a, (96, 64)), (gs_b, (320, 200))):     # images are allocations: made before the measurement starts
            loop = hip.FrameLoop(ctx, gs, w, h, "compute")
            loop.renderpass.execute = (lambda lp: lambda: ctx.dispatch_static("compute", lp.renderpass.constants, *lp.renderpass.group_count))(loop)
            loops.append(loop)
        ctx.sync()
        before = free_device_memory()
        for _ in range(3):
            for loop in loops:
                loop.frames(2)
        ctx.sync()
        assert free_device_memory() == before
        want, _, _ = oracle_frames(b, 6, use_flat=True, materials=gs_b.materials_host, textures=gs_b.texture_table)
        assert np.array_equal(loops[1].latest().read(), want)

        # scene metadata in a buffer the library keeps no host copy of: reading it would mean a copy back and a wait
        big_meta = ctx.create_buffer(8192)
        big_meta.write(gs_a.meta, 0, 160)
        rp = loops[0].renderpass
        rp.set_constant(0, big_meta)
        with pytest.raises(hip.HipError, match="no host copy"):
            ctx.dispatch_static("compute", rp.constants, *rp.group_count)
        ctx.dispatch("compute", rp.constants, *rp.group_count)        # the allocating entry point reads it back and renders
        rp.set_constant(0, gs_a.scene_buffer)
        # a layout whose header no dispatch has read yet
        fresh = hip.GpuScene(ctx, a, a["nodes"], a["flat"], a["meta"])
        ctx.set_option("scene.validate", 0)
        try:
            lp = hip.FrameLoop(ctx, fresh, 96, 64, "compute")
            with pytest.raises(hip.HipError, match="header of the flattened layout"):
                lp.renderpass.execute = lambda: ctx.dispatch_static("compute", lp.renderpass.constants, *lp.renderpass.group_count)
                lp.frame()
        finally:
            ctx.set_option("scene.validate", 1)
    finally:
        ctx.close()
