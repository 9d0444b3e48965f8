#!/usr/bin/env python3
"""How much would overlapping consecutive frames buy?  Renders the benchmark scene from TWO contexts (two streams)
whose dispatches are interleaved from one host thread, against one context doing the same number of frames."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd", "python"))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--triangles", type=int, default=2_800_000)
    ap.add_argument("--scene", default="street_exterior")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--texture-size", type=int, default=512)
    ap.add_argument("--contexts", type=int, default=2)
    ap.add_argument("--in-flight", type=int, default=1, help="frames.in_flight of every context")
    args = ap.parse_args()
    import bench
    from path_tracer_amd import hip
    sc, nodes, flat, meta, _ = bench.build_scene(args, 0)
    loops, ctxs = [], []
    for _ in range(args.contexts):
        ctx = hip.Context(0)
        ctx.set_option("frames.in_flight", args.in_flight)
        gpu = hip.GpuScene(ctx, sc, nodes, flat, meta)
        loops.append(hip.FrameLoop(ctx, gpu, args.width, args.height, "compute"))
        ctxs.append(ctx)
    for n_active in (1, args.contexts):
        for rep in range(3):
            for l in loops[:n_active]:
                for _ in range(3):
                    l.frame()
            for c in ctxs:
                c.sync()
            t0 = time.perf_counter()
            for _ in range(args.frames):
                for l in loops[:n_active]:
                    l.frame()
            for c in ctxs:
                c.sync()
            dt = time.perf_counter() - t0
            print(f"{n_active} context(s): {dt / (args.frames * n_active) * 1e3:.4f} ms per frame", flush=True)


if __name__ == "__main__":
    main()
