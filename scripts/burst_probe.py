#!/usr/bin/env python3
"""Does the frame time depend on how many frames are queued back to back?  Renders chunks of frames with a sync after each
chunk and prints every chunk's ms per frame: a slowdown that builds up across chunks is the device (clocks, power), one
that depends on the chunk length alone is queueing."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd", "python"))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, nargs="+", default=[20] * 10 + [100, 100, 20, 20])
    ap.add_argument("--pause", type=float, default=0.0, help="seconds of idle time before each chunk")
    ap.add_argument("--triangles", type=int, default=2_800_000)
    ap.add_argument("--scene", default="street_exterior")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--texture-size", type=int, default=512)
    ap.add_argument("--instances", default="1x1")
    ap.add_argument("--two-level", action="store_true")
    ap.add_argument("--bvh", default="host", choices=["host", "gpu"])
    ap.add_argument("--in-flight", type=int, default=2)
    ap.add_argument("--reset", action="store_true", help="restart the sample index with every chunk")
    ap.add_argument("--per-dispatch", type=int, default=1, help="samples per dispatch (1 = the reference's protocol)")
    ap.add_argument("--summary", action="store_true", help="print the distribution of the chunks' times instead of every chunk (long soaks)")
    ap.add_argument("--per-sample", type=int, default=0, help="instead: print rays, nodes per ray and ms of each of the first N sample indices")
    args = ap.parse_args()
    import bench
    from path_tracer_amd import hip
    sc, nodes, flat, meta, _ = bench.build_scene(args, 0)
    ctx = hip.Context(0)
    gpu = hip.GpuScene(ctx, sc, nodes, flat, meta)
    if args.bvh == "gpu":
        gpu.wire(ctx.create_compute_renderpass())
        ctx.bind_flat(gpu.bvh_buffer, ctx.build_flat(gpu.indices_buffer, gpu.positions_buffer, sc["triangles"].shape[0]))
    loop = hip.FrameLoop(ctx, gpu, args.width, args.height, "compute", samples_per_dispatch=args.per_dispatch)
    ctx.set_option("frames.in_flight", args.in_flight)
    loop.frame(); ctx.sync()
    if args.per_sample:
        loop.reset()
        ctx.set_option("frames.in_flight", 1)
        for k in range(1, args.per_sample + 1):
            ctx.stats_read(reset=True); ctx.stats_enable(True)
            loop.frame(); ctx.sync()
            st = ctx.stats_read(reset=True); ctx.stats_enable(False)
            loop.sample_index -= 1                      # the same sample again, timed without the counters
            t0 = time.perf_counter()
            loop.frame(); ctx.sync()
            ms = (time.perf_counter() - t0) * 1e3
            print(f"sample {k}: {ms:.3f} ms rays {st['rays']} nodes/ray {st['nodes'] / max(1, st['rays']):.2f} triangles/ray {st['triangles'] / max(1, st['rays']):.2f}", flush=True)
        ctx.close()
        return
    out = []
    for n in args.chunks:
        if args.reset:
            loop.reset()
        if args.pause:
            time.sleep(args.pause)
        t0 = time.perf_counter()
        if args.per_dispatch > 1:
            loop.frames(n)
        else:
            for _ in range(n):
                loop.frame()
        ctx.sync()
        out.append((n, round((time.perf_counter() - t0) / n * 1e3, 3)))
        if args.summary and len(out) % 500 == 0:
            print(f"... {len(out)} chunks", flush=True)
    if args.summary:
        ms = sorted(m for _, m in out)
        pick = lambda q: ms[min(len(ms) - 1, int(q * len(ms)))]
        print(f"{sum(n for n, _ in out)} steps in {len(out)} chunks, {args.per_dispatch} sample(s) per dispatch: ms per step median {pick(0.5):.3f}, "
              f"90th percentile {pick(0.9):.3f}, 99th {pick(0.99):.3f}, worst chunk {ms[-1]:.3f}, best {ms[0]:.3f}", flush=True)
    else:
        print(" ".join(f"{n}:{ms}" for n, ms in out), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
