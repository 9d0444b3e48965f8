#!/usr/bin/env python3
"""What one rank of an N-GPU tile partition costs, measured on ONE GPU (no exchange): K steps of the bench scene with
ptr_set_partition(rank 0 of N), for several dispatch sizes.  T_1 / (N * T_N) of these figures is the scaling efficiency the
render part alone allows; the all-gather and the barriers come on top.
   python scripts/partition_probe.py --worlds 1 2 4 8 --steps 20"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd", "python"))
sys.path.insert(0, ROOT)


def main():
    import bench
    ap = bench.make_parser()
    ap.add_argument("--worlds", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--per-dispatch", type=int, nargs="+", default=[10, 16, 20])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--schedules", nargs="*", default=[], help='explicit dispatch sizes to try as well, e.g. "13,7" "12,8" (each must add up to --steps)')
    args = ap.parse_args()
    from path_tracer_amd import hip
    sc, nodes, flat, meta, _ = bench.build_scene(args, 0)
    ctx = hip.Context(0)
    gpu = hip.GpuScene(ctx, sc, nodes, flat, meta)
    def measure(world, per, schedule=None):
        ctx.set_partition(0, world, args.stripe_rows)
        loop = hip.FrameLoop(ctx, gpu, args.width, args.height, args.pipeline, samples_per_dispatch=min(64, per))
        loop.frames(2 * min(64, per)); ctx.sync()   # both frame slots sized for the largest dispatch
        best = 1e9
        for _ in range(args.rounds):
            loop.reset()
            t0 = time.perf_counter()
            loop.frames(args.steps, schedule)
            ctx.sync()
            best = min(best, time.perf_counter() - t0)
        return best

    # T_1: the whole frames on one GPU, with the dispatch size that suits ONE GPU best (not the one the partition is tried with)
    t1 = {per: measure(1, per) for per in args.per_dispatch}
    for per, t in t1.items():
        print(f"world 1  samples per dispatch {min(64, per):2d}: {t * 1e3:7.3f} ms for {args.steps} steps", flush=True)
    t1_best = min(t1.values())
    for world in [w for w in args.worlds if w > 1]:
        for per in args.per_dispatch:
            t = measure(world, per)
            print(f"world {world}  samples per dispatch {min(64, per):2d}: {t * 1e3:7.3f} ms for {args.steps} steps"
                  f"   render-only efficiency {t1_best / (world * t):.3f} (against the best one-GPU time, {t1_best * 1e3:.3f} ms)", flush=True)
    for text in args.schedules:
        schedule = [int(v) for v in text.split(",")]
        for world in args.worlds:
            t = measure(world, max(schedule), schedule)
            print(f"world {world}  dispatches of {text}: {t * 1e3:7.3f} ms for {args.steps} steps   render-only efficiency {t1_best / (world * t):.3f}", flush=True)
    ctx.set_partition(0, 1, 8)
    ctx.close()


if __name__ == "__main__":
    main()
