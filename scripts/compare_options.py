#!/usr/bin/env python3
"""Renders the same frames under several option sets and compares the images bit for bit (first set = the reference).
   python scripts/compare_options.py --configs "pipeline.fused=0" "pipeline.fused=1" [--frames 3] [scene args as bench.py]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd", "python"))
sys.path.insert(0, ROOT)


def main():
    import bench
    ap = bench.make_parser()
    ap.add_argument("--configs", nargs="+", required=True)
    ap.add_argument("--frames", type=int, default=3)
    args = ap.parse_args()
    from path_tracer_amd import hip
    sc, nodes, flat, meta, _ = bench.build_scene(args, 0)
    ctx = hip.Context(0)
    gpu = hip.GpuScene(ctx, sc, nodes, flat, meta)
    images = []
    for cfg in args.configs:
        for kv in filter(None, cfg.split(",")):
            ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        gpu.set_meta(meta)
        loop = hip.FrameLoop(ctx, gpu, args.width, args.height, args.pipeline)
        for _ in range(args.frames):
            loop.frame()
        images.append(loop.latest().read())
        bad = int((images[-1] != images[0]).any(axis=-1).sum())
        print(f"{cfg}: {bad} of {args.width * args.height} pixels differ from '{args.configs[0]}'", flush=True)
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
