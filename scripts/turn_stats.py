#!/usr/bin/env python3
"""Wave-level step counts of the traversal kernel's counting variant on a bench scene: rays, nodes and triangles per ray, inner
and leaf turns of the waves, lanes with a ray per inner step.  usage: turn_stats.py [bench.py scene arguments]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd", "python")); sys.path.insert(0, ROOT)
import bench
from path_tracer_amd import hip
args = bench.make_parser().parse_args(sys.argv[1:])
sc, nodes, flat, meta, _ = bench.build_scene(args, 0)
ctx = hip.Context(0)
gpu = hip.GpuScene(ctx, sc, nodes, flat, meta)
loop = hip.FrameLoop(ctx, gpu, args.width, args.height, "compute", samples_per_dispatch=10)
ctx.stats_read(reset=True); ctx.stats_enable(True)
loop.frames(10); ctx.sync()
s = ctx.stats_read(reset=True); ctx.stats_enable(False)
r = s["rays"]
print({k: v for k, v in s.items()})
print(f"nodes/ray {s['nodes'] / r:.2f} tris/ray {s['triangles'] / r:.2f}; inner steps {s['wave_inner_steps']} leaf turns {s['wave_leaf_steps']}: "
      f"inner-step lane use {s['nodes'] / (64.0 * s['wave_inner_steps']):.3f}, lanes without a ray per inner step {s['idle_lane_steps'] / (64.0 * s['wave_inner_steps']):.3f}, "
      f"triangles per leaf turn and lane {s['triangles'] / (64.0 * s['wave_leaf_steps']):.3f}, inner steps per ray-wave {s['wave_inner_steps'] * 64.0 / r:.2f}, leaf turns x64 per ray {s['wave_leaf_steps'] * 64.0 / r:.2f}")
ctx.close()
