#!/usr/bin/env python3
"""A/B timing of tuning knobs in ONE process (interleaved rounds, same scene, same GPU):
   python scripts/ab.py --configs "extend.refill=16" "extend.refill=32" "extend.refill=16,extend.blocks_per_cu=7"
(options persist from one configuration to the next: name every key in every configuration)
Prints per-config median / min ms per frame and the per-kernel split from the library's HIP-event profiler."""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd", "python"))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", nargs="+", default=["extend.refill=16", "extend.refill=32"])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--triangles", type=int, default=2_800_000)
    ap.add_argument("--scene", default="street_exterior")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--texture-size", type=int, default=512)
    ap.add_argument("--pipeline", default="compute")
    ap.add_argument("--instances", default="1x1")
    ap.add_argument("--two-level", action="store_true")
    ap.add_argument("--bvh", default="host", choices=["host", "gpu"])
    ap.add_argument("--no-profile", action="store_true", help="time whole frames only (no events between kernels)")
    args = ap.parse_args()
    import bench
    from path_tracer_amd import hip
    sc = None
    if os.environ.get("PTR_AB_TORCH"):       # is the frame overlap the same in a process that also runs torch?
        import torch
        keep = torch.zeros((args.height, args.width, 4), dtype=torch.float32, device="cuda")
        if os.environ["PTR_AB_TORCH"] == "stream":
            keep_stream = torch.cuda.Stream()
        torch.cuda.synchronize()
    sc, nodes, flat, meta, _ = bench.build_scene(args, 0)
    ctx = hip.Context(0)
    gpu = hip.GpuScene(ctx, sc, nodes, flat, meta)
    if args.bvh == "gpu":
        gpu.wire(ctx.create_compute_renderpass())
        ctx.bind_flat(gpu.bvh_buffer, ctx.build_flat(gpu.indices_buffer, gpu.positions_buffer, sc["triangles"].shape[0]))
    textures = None
    if os.environ.get("PTR_AB_TORCH") == "wrap":  # the two images live in torch tensors, as in bench.py
        ia = torch.zeros((args.height, args.width, 4), dtype=torch.float32, device="cuda")
        ib = torch.zeros_like(ia)
        textures = (ctx.wrap_2d_texture(ia.data_ptr(), args.width, args.height), ctx.wrap_2d_texture(ib.data_ptr(), args.width, args.height))
    loop = hip.FrameLoop(ctx, gpu, args.width, args.height, args.pipeline, textures=textures)
    configs = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in c.split(",") if kv) for c in args.configs]
    times = [[] for _ in configs]
    kernels = [{} for _ in configs]
    for rnd in range(args.rounds + 1):
        for i, cfg in enumerate(configs):
            for k, v in cfg.items():
                if k == "dispatch.samples":              # the frame loop sets this one per dispatch
                    loop.samples_per_dispatch = v
                else:
                    ctx.set_option(k, v)
            loop.reset()
            loop.frames(loop.samples_per_dispatch); ctx.sync()   # settle (scratch re-allocation, clocks)
            ctx.profile_read(reset=True); ctx.profile_enable(not args.no_profile)
            t0 = time.perf_counter()
            loop.frames(args.frames)
            ctx.sync()
            dt = (time.perf_counter() - t0) / args.frames * 1e3
            ctx.profile_enable(False)
            prof = ctx.profile_read(reset=True)
            if rnd == 0:
                continue                                 # first round is warm-up
            times[i].append(dt)
            for n, v in prof.items():
                kernels[i].setdefault(n, []).append(v["total_ms"] / args.frames)
    for i, cfg in enumerate(configs):
        print(json.dumps(dict(config=args.configs[i], median_ms=round(statistics.median(times[i]), 4), min_ms=round(min(times[i]), 4),
                              kernels_ms={n: round(statistics.median(v), 4) for n, v in sorted(kernels[i].items())})), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
