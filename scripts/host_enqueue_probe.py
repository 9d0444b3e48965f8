import os, sys, time
ROOT='/root/repo'
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd", "python")); sys.path.insert(0, ROOT)
import bench
from path_tracer_amd import hip
args = bench.make_parser().parse_args([])
sc, nodes, flat, meta, _ = bench.build_scene(args, 0)
ctx = hip.Context(0)
gpu = hip.GpuScene(ctx, sc, nodes, flat, meta)
for world in (1, 8):
    ctx.set_partition(0, world, 8)
    loop = hip.FrameLoop(ctx, gpu, 1920, 1080, "compute", samples_per_dispatch=10)
    loop.frames(20); ctx.sync()
    for rnd in range(4):
        loop.reset()
        t0 = time.perf_counter(); loop.frame(10); t1 = time.perf_counter(); loop.frame(10); t2 = time.perf_counter(); ctx.sync(); t3 = time.perf_counter()
        print(f"world {world}: first dispatch enqueued in {(t1-t0)*1e6:.0f} us, second in {(t2-t1)*1e6:.0f} us, all done after {(t3-t0)*1e6:.0f} us", flush=True)
    # pieces of one frame() call
    t0 = time.perf_counter()
    for _ in range(100): ctx.set_option("dispatch.samples", 10)
    t1 = time.perf_counter()
    for _ in range(100): gpu.update_meta(sample_index=1)
    t2 = time.perf_counter()
    for _ in range(100): loop.renderpass.set_ouput_texture(loop.output); loop.renderpass.set_constant(60, loop.accumulation)
    t3 = time.perf_counter(); ctx.sync()
    print(f"  set_option {(t1-t0)*1e4:.1f} us, update_meta {(t2-t1)*1e4:.1f} us, two constants {(t3-t2)*1e4:.1f} us", flush=True)
ctx.set_partition(0, 1, 8)
ctx.close()
