#!/usr/bin/env python3
"""Lane utilisation of the traversal kernel per bounce count: runs frames with the counting kernel variants."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd", "python")); sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--triangles", type=int, default=2_800_000); ap.add_argument("--scene", default="street_exterior")
ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--texture-size", type=int, default=512); ap.add_argument("--bounces", type=int, default=4)
ap.add_argument("--set", nargs="*", default=[], help="library options, key=value")
ap.add_argument("--samples-per-dispatch", type=int, default=1)
ap.add_argument("--instances", default="1x1"); ap.add_argument("--two-level", action="store_true")
args = ap.parse_args()
import bench
from path_tracer_amd import hip
sc, nodes, flat, meta, _ = bench.build_scene(args, 0)
ctx = hip.Context(0)
for kv in args.set:
    ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
gpu = hip.GpuScene(ctx, sc, nodes, flat, meta)
loop = hip.FrameLoop(ctx, gpu, args.width, args.height, "compute")
prev = None
for mb in range(1, args.bounces + 1):
    gpu.update_meta(max_bounce=mb)
    loop.reset(); ctx.stats_read(reset=True); ctx.stats_enable(True)
    loop.frame(args.samples_per_dispatch)
    st = ctx.stats_read(reset=True); ctx.stats_enable(False)
    cur = st
    if prev is not None:
        st = {k: cur[k] - prev[k] for k in cur}
    prev = cur
    print(f"bounce {mb - 1}: rays {st['rays']} nodes/ray {st['nodes'] / max(1, st['rays']):.1f} tris/ray {st['triangles'] / max(1, st['rays']):.2f} "
          f"inner util {st['nodes'] / max(1, 64 * st['wave_inner_steps']):.3f} (lanes without a ray {st['idle_lane_steps'] / max(1, 64 * st['wave_inner_steps']):.3f}) "
          f"leaf turns {st['wave_leaf_steps']}", flush=True)
ctx.close()
