#!/usr/bin/env python3
"""Wave-level step counts of the traversal kernel's counting variant on a bench scene: rays, nodes and triangles per ray, inner
and leaf turns of the waves, and where the lanes of an inner step are -- at an inner node, without a ray, or waiting at a leaf.
With --by-bounce the same per bounce: the counters of a run with k bounces minus those of a run with k - 1 (the rays of bounce
k - 1 are the same in both; the counting variant always looks for closest hits).
   usage: turn_stats.py [--by-bounce] [bench.py scene arguments]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd", "python")); sys.path.insert(0, ROOT)
import bench
from path_tracer_amd import hip

by_bounce = "--by-bounce" in sys.argv
args = bench.make_parser().parse_args([a for a in sys.argv[1:] if a != "--by-bounce"])
sc, nodes, flat, meta, _ = bench.build_scene(args, 0)
ctx = hip.Context(0)
gpu = hip.GpuScene(ctx, sc, nodes, flat, meta)


def run(bounces):
    gpu.update_meta(max_bounce=bounces)
    loop = hip.FrameLoop(ctx, gpu, args.width, args.height, "compute", samples_per_dispatch=10)
    ctx.stats_read(reset=True); ctx.stats_enable(True)
    loop.frames(10); ctx.sync()
    s = ctx.stats_read(reset=True); ctx.stats_enable(False)
    for t in (loop.accumulation, loop.output):
        t.destroy()
    return s


def describe(s, what):
    r = max(1, s["rays"])
    inner = max(1, s["wave_inner_steps"])
    use, none = s["nodes"] / (64.0 * inner), s["idle_lane_steps"] / (64.0 * inner)
    print(f"{what}: rays {s['rays']}  nodes/ray {s['nodes'] / r:.2f}  triangles/ray {s['triangles'] / r:.2f}  inner steps {s['wave_inner_steps']}  leaf turns {s['wave_leaf_steps']}\n"
          f"    lanes of an inner step: at an inner node {use:.3f}, without a ray {none:.3f}, waiting at a leaf {1.0 - use - none:.3f};  "
          f"triangles per leaf turn and lane {s['triangles'] / (64.0 * max(1, s['wave_leaf_steps'])):.3f};  inner steps x 64 per ray {inner * 64.0 / r:.2f}, leaf turns x 64 per ray {s['wave_leaf_steps'] * 64.0 / r:.2f}")


total = run(args.bounces)
print({k: v for k, v in total.items()})
describe(total, f"all {args.bounces} bounces, 10 samples")
if by_bounce:
    before = {k: 0 for k in total}
    for k in range(1, args.bounces + 1):
        now = total if k == args.bounces else run(k)
        describe({key: now[key] - before[key] for key in now}, f"bounce {k - 1}")
        before = now
ctx.close()
