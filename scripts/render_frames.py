#!/usr/bin/env python3
"""Minimal driver for profilers: builds the benchmark scene and renders N frames of the "compute" dispatch.
   rocprofv3 --kernel-trace --stats -d out -- python3 scripts/render_frames.py --frames 20"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd", "python"))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--triangles", type=int, default=2_800_000)
    ap.add_argument("--scene", default="street_exterior")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--texture-size", type=int, default=512)
    ap.add_argument("--pipeline", default="compute")
    ap.add_argument("--option", action="append", default=[])
    ap.add_argument("--samples-per-dispatch", type=int, default=1)
    args = ap.parse_args()
    import bench
    from path_tracer_amd import hip
    sc, nodes, flat, meta, _ = bench.build_scene(args, 0)
    ctx = hip.Context(0)
    for kv in args.option:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    gpu = hip.GpuScene(ctx, sc, nodes, flat, meta)
    loop = hip.FrameLoop(ctx, gpu, args.width, args.height, args.pipeline, samples_per_dispatch=args.samples_per_dispatch)
    loop.frames(args.frames)
    ctx.sync()
    print("rendered", args.frames, "frames; mean of last image", float(loop.latest().read()[..., :3].mean()))
    ctx.close()


if __name__ == "__main__":
    main()
