#!/usr/bin/env python3
"""Headline benchmark of the render hot path: Mrays/s and ms/frame, 1920x1080, 1 sample per step, on a
Bistro-Exterior-class synthetic scene (the reference's real assets are Git-LFS pointers, unavailable offline).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one full-resolution 1-spp pass of the "compute" dispatch (generate + per-bounce extend/shade +
accumulate), inputs resident in HBM.  With N GPUs the scene is replicated and the SAMPLES are partitioned
(rank r draws the global samples r+1, r+1+N, ...), one RCCL all-reduce of the linearized images closes the
timed region: per-GPU work is fixed, so the scaling is weak.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd", "python"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def build_scene(args, rank):
    from path_tracer_amd import host, scenes
    t0 = time.time()
    sc = scenes.generate(args.scene, args.triangles, texture_size=args.texture_size)
    instanced = getattr(args, "instances", "1x1") not in ("", "1x1")
    if instanced:
        # the 10^8-triangle class: translated copies of the scene on a grid, one flat scene in the 4-word index layout.
        # The reference's single-threaded builder and its 2N-1 node array (minutes, 13 GB at 10^8 triangles) are skipped:
        # the wavefront path never reads the threaded tree; a one-leaf placeholder stands in for the node buffer.
        nx, nz = (int(v) for v in args.instances.lower().split("x"))
        two_level = bool(getattr(args, "two_level", False))
        if two_level:
            # the same copies as placements of ONE object: its tree, and a top level over the copies' boxes
            sc = dict(sc, instance_transforms=scenes.grid_transforms(sc, nx, nz), copies=nx * nz,
                      kind=f"{sc.get('kind', 'scene')} x{nx * nz} (two-level)")
        else:
            sc = scenes.instance_grid(sc, nx, nz)
        t1 = time.time()
        nodes = np.zeros((1, 8), np.uint32)
        nodes[0, 3] = 0xFFFFFFFF
        flat = None if getattr(args, "bvh", "host") == "gpu" else host.bvh_flatten(sc["triangles"], None, threads=min(16, os.cpu_count() or 1))
        if two_level and flat is not None:
            sc["flat_instances"] = host.bvh_flatten_instances(flat, sc["instance_transforms"])
        t3 = time.time()
        c = sc["camera"]
        cam = host.camera(c["pos"], c["target"], c["v_fov"], args.width / args.height, c["aperture"], c["focus"])
        meta = scenes.pack_metadata(cam, args.width, args.height, max_bounce=args.bounces, min_bounce=1)
        log(rank, f"scene {sc['kind']}: {sc['triangles'].shape[0]} triangles; generate {t1 - t0:.1f}s, flattened BVH {t3 - t1:.1f}s")
        return sc, nodes, flat, meta, dict(generate_s=t1 - t0, threaded_bvh_s=0.0, flatten_s=t3 - t1, threaded_mode="none (placeholder)")
    t1 = time.time()
    try:
        nodes = host.bvh_build(sc["triangles"], host.REFERENCE_ORDER)
        ranks = host.bvh_visit_ranks(nodes, sc["triangles"].shape[0])
        ref_note = "reference_order"
    except host.HostError as e:  # the reference's builder cannot finish on this input: ties fall back to file order
        log(rank, "reference_order build failed:", e)
        nodes = host.bvh_build(sc["triangles"], host.SAH_PARTITIONED)
        ranks, ref_note = None, "sah_partitioned (reference_order does not terminate)"
    t2 = time.time()
    flat = host.bvh_flatten(sc["triangles"], ranks, threads=min(16, os.cpu_count() or 1))
    t3 = time.time()
    c = sc["camera"]
    cam = host.camera(c["pos"], c["target"], c["v_fov"], args.width / args.height, c["aperture"], c["focus"])
    meta = scenes.pack_metadata(cam, args.width, args.height, max_bounce=args.bounces, min_bounce=1)
    log(rank, f"scene {args.scene}: {sc['triangles'].shape[0]} triangles, {sc['primitive_count']} primitives; generate {t1 - t0:.1f}s, "
              f"threaded BVH ({ref_note}) {t2 - t1:.1f}s, flattened BVH {t3 - t2:.1f}s")
    return sc, nodes, flat, meta, dict(generate_s=t1 - t0, threaded_bvh_s=t2 - t1, flatten_s=t3 - t2, threaded_mode=ref_note)


def cpu_baseline(args, sc, nodes, flat, meta, gpu_scene, budget_s):
    """The oracle (a scalar C port of the reference shader, walking the reference-order tree exactly like the
    reference) on the host cores, on a bounded sample of the same workload: bands of rows spread over the frame."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    pyoracle.build()
    threads = min(os.cpu_count() or 1, 16)
    arrays = dict(sc, meta=meta, nodes=nodes, flat=flat, materials=gpu_scene.materials_host, textures=gpu_scene.texture_table)
    o = pyoracle.OracleScene(arrays)
    o.set_meta(sample_index=1)
    use_flat = nodes.shape[0] < 2 * sc["triangles"].shape[0] - 1     # no threaded tree was built: walk the flattened layout
    H, W = args.height, args.width
    acc = np.zeros((H, W, 4), np.float32)
    out = np.zeros_like(acc)
    band = 4 * threads                          # the oracle's workers take rows of a call one at a time: four rows each
    starts = list(range(0, H - band + 1, band))
    order = [starts[(i * 37) % len(starts)] for i in range(len(starts))]  # spread over the frame
    rays = 0
    rows_done = 0
    t0 = time.time()
    sample_index = 1
    while time.time() - t0 < budget_s and sample_index <= 64:       # whole frames, sample after sample, until the budget is used
        o.set_meta(sample_index=sample_index)
        for y in order:
            c = o.render(acc, out, rows=(y, y + band), threads=threads, use_flat=use_flat)
            rays += c["rays"]
            rows_done += band
            if time.time() - t0 >= budget_s:
                break
        acc, out = out, acc
        sample_index += 1
    dt = time.time() - t0
    o.set_meta(sample_index=1)
    # one thread on a few of the same bands, for the per-core figure
    single_rays, t1 = 0, time.time()
    for y in order[:64]:
        single_rays += o.render(acc, out, rows=(y, y + band), threads=1, use_flat=use_flat)["rays"]
        if time.time() - t1 >= max(1.0, budget_s * 0.25):
            break
    single_dt = time.time() - t1
    return dict(value=rays / dt / 1e6, unit="Mrays/s", cores=threads, kind="port", single_thread_value=single_rays / single_dt / 1e6,
                sample=f"{rows_done} rows of {H}-row frames ({rows_done // band} bands of {band} rows spread over the frame, sample_index 1..{sample_index - 1}), "
                       f"{rays} rays in {dt:.1f}s, " + ("flattened BVH" if use_flat else "reference-order threaded BVH"))


def lavapipe_probe():
    """The reference's own shader could only be timed on the host through a CPU Vulkan driver (Mesa lavapipe) plus a GLSL
    compiler; neither ships in this image.  Reported so that the baseline line says what it is and what it is not."""
    import ctypes.util
    import glob
    import shutil
    missing = []
    if not (shutil.which("glslc") or shutil.which("glslangValidator")):
        missing.append("no GLSL compiler (glslc / glslangValidator)")
    if not ctypes.util.find_library("vulkan"):
        missing.append("no libvulkan")
    if not glob.glob("/usr/share/vulkan/icd.d/*lvp*") and not glob.glob("/etc/vulkan/icd.d/*lvp*"):
        missing.append("no lavapipe ICD")
    return "available (not driven by this benchmark)" if not missing else "unavailable: " + ", ".join(missing)


def pmc_traffic(kernel, workload):
    """HBM bytes per launch of `kernel` from the committed counter passes of this same command (profiles/traffic.json,
    written by scripts/traffic_from_pmc.py): FETCH_SIZE and WRITE_SIZE come from separate rocprofv3 --pmc runs, so
    they cannot be collected inside this process.  None when the file is missing or was measured on another workload."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None, "no profiles/traffic.json", {}
    rec = t.get("kernels", {}).get(kernel)
    if rec is None or t.get("workload") != workload:
        return None, "profiles/traffic.json holds no counters for this kernel and workload", {}
    extra = {k: rec[k] for k in ("valu_busy", "l2_hit_rate", "l1_hit_rate", "waves_waiting") if k in rec}
    return rec["bytes_per_launch"], t.get("note", ""), extra


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scene", default="street_exterior")
    ap.add_argument("--triangles", type=int, default=2_800_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--texture-size", type=int, default=512)
    ap.add_argument("--instances", default="1x1", help="NXxNZ translated copies of the scene on a grid (e.g. 6x6 of the default scene = 10^8 triangles)")
    ap.add_argument("--two-level", action="store_true", help="with --instances: keep ONE copy of the scene and its tree, and place it through a top-level tree")
    ap.add_argument("--pipeline", default="compute", choices=["compute", "compute.megakernel", "compute.wavefront"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--bvh", default="host", choices=["host", "gpu"], help="host: binned-SAH flattener (default); gpu: ptr_bvh_build_flat")
    ap.add_argument("--stream", default="torch", choices=["torch", "own"], help="the context's stream: torch's current stream or its own")
    ap.add_argument("--frames-in-flight", type=int, default=2, help="consecutive frames overlapped on internal streams (1 = none)")
    args = ap.parse_args()

    # Rank 0 must print ONE line on stdout.  Libraries underneath write there too (RCCL prints a version banner when the
    # first communicator is created): everything written to fd 1 from here on goes to stderr, the result line goes to the
    # saved descriptor.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(rank, f"note: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    # PTR_BENCH_DEVICE / PTR_BENCH_BACKEND exist to rehearse the multi-rank path on a one-GPU box
    # (all ranks on one device, gloo instead of RCCL); the driver's runs use neither.
    if "PTR_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["PTR_BENCH_DEVICE"])
    backend = os.environ.get("PTR_BENCH_BACKEND", "nccl")
    # PTR_BENCH_FORCE_DIST=1: run the exchange path (process group, barriers, linearize / all-reduce / resolve) with ONE rank,
    # to rehearse the RCCL leg on a one-GPU box
    multi = world > 1 or bool(os.environ.get("PTR_BENCH_FORCE_DIST"))
    torch.cuda.set_device(local_rank)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from path_tracer_amd import distributed as ptdist
    from path_tracer_amd import hip, host

    sc, nodes, flat, meta, build_times = build_scene(args, rank)

    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx = hip.Context(local_rank, stream=stream.cuda_stream if args.stream == "torch" else None)
        for kv in filter(None, os.environ.get("PTR_BENCH_OPTIONS", "").split(",")):   # tuning experiments: "key=value,key=value"
            ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        t_up = time.perf_counter()
        gpu = hip.GpuScene(ctx, sc, nodes, flat, meta)
        ctx.sync()
        # the one-time host -> device copy of the scene (pageable host memory): never part of the timed region
        build_times["upload_s"] = time.perf_counter() - t_up
        build_times["upload_mb"] = sum(int(np.asarray(sc[k]).nbytes) for k in ("indices", "positions", "normals", "uvs")) / 1e6 \
            + (0 if flat is None else flat.nbytes / 1e6) + nodes.nbytes / 1e6 + sum(t.nbytes for t in sc["textures"] if t is not None) / 1e6
        if args.bvh == "gpu":
            gpu.wire(ctx.create_compute_renderpass())          # the context's index layout for this scene
            ctx.profile_enable(True)
            t_build = time.perf_counter()
            gpu_flat = ctx.build_flat(gpu.indices_buffer, gpu.positions_buffer, sc["triangles"].shape[0])
            build_times["gpu_build_wall_s"] = time.perf_counter() - t_build
            ctx.profile_enable(False)
            build_times["gpu_build_device_ms"] = ctx.profile_read(reset=True).get("bvh.build_flat", dict(total_ms=0.0))["total_ms"]
            ctx.bind_flat(gpu.bvh_buffer, gpu_flat)
            if sc.get("instance_transforms") is not None:   # the top level needs the object's header only: bounds, root, stack need
                sc["flat_instances"] = host.bvh_flatten_instances(gpu_flat.read(np.uint8, 0, 64), sc["instance_transforms"])
                gpu.instance_buffer = ctx.create_buffer(sc["flat_instances"].nbytes)
                gpu.instance_buffer.write(sc["flat_instances"])
                ctx.bind_instances(gpu_flat, gpu.instance_buffer)
            flat = gpu_flat.read(np.uint8) if args.cpu_seconds > 0 else flat
            log(rank, f"GPU BVH build: {build_times['gpu_build_device_ms']:.1f} ms on the device, {build_times['gpu_build_wall_s'] * 1e3:.1f} ms wall")
        W, H = args.width, args.height
        img_a = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        img_b = torch.zeros_like(img_a)
        tex_a = ctx.wrap_2d_texture(img_a.data_ptr(), W, H)
        tex_b = ctx.wrap_2d_texture(img_b.data_ptr(), W, H)
        by_handle = {tex_a.handle: img_a, tex_b.handle: img_b}
        loop = hip.FrameLoop(ctx, gpu, W, H, args.pipeline, textures=(tex_a, tex_b))
        if world > 1:
            ctx.set_sample_partition(rank, world)
        ctx.set_option("frames.in_flight", args.frames_in_flight)

        exchange = ctx.create_compute_renderpass()

        def combine():
            """sample partition: linearize -> all-reduce -> resolve; returns the final image tensor"""
            latest, scratch = loop.latest(), loop.output
            exchange.set_pipeline("linearize")
            exchange.set_ouput_texture(scratch)
            exchange.set_constant(60, latest)
            exchange.set_dispatch_size(1, 1, 1)
            exchange.execute()
            ptdist.allreduce_linear(by_handle[scratch.handle])
            exchange.set_pipeline("resolve")
            exchange.set_ouput_texture(latest)
            exchange.set_constant(60, scratch)
            exchange.execute()
            return by_handle[latest.handle]

        def barrier():
            if multi:
                dist.barrier()
            torch.cuda.synchronize()

        for _ in range(args.warmup):
            loop.frame()
        if multi:
            combine()
        barrier()
        loop.reset()
        img_a.zero_(); img_b.zero_()
        # live kernel timing inside the timed region: HIP events around the dominant kernel, on the stream it runs on
        ctx.profile_read(reset=True)
        ctx.set_option("profile.extend_only", 1)
        ctx.profile_enable(not os.environ.get("PTR_BENCH_NO_LIVE_PROFILE"))
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loop.frame()
        if multi:
            combine()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        ctx.profile_enable(False)
        prof = ctx.profile_read(reset=True)
        ctx.set_option("profile.extend_only", 0)

        # what a caller pays on top when it reads every finished frame back over PCIe (33 MB at 1080p)
        t_rb = time.perf_counter()
        for _ in range(3):
            loop.latest().read()
        build_times["frame_readback_ms"] = (time.perf_counter() - t_rb) / 3 * 1e3

        # per-kernel split of a frame that has the GPU to itself (one frame in flight), outside the timed region
        loop.reset()
        ctx.set_option("frames.in_flight", 1)
        ctx.profile_enable(True)
        solo_frames = args.steps              # the same sample indices as the timed frames: a frame's cost depends on its sample index
        for _ in range(solo_frames):
            loop.frame()
        torch.cuda.synchronize()
        ctx.profile_enable(False)
        solo = ctx.profile_read(reset=True)
        ctx.set_option("frames.in_flight", args.frames_in_flight)

        # exact work of the timed frames: re-run the same sample indices with the counting kernel variants
        loop.reset()
        ctx.stats_read(reset=True)
        ctx.stats_enable(True)
        for _ in range(args.steps):
            loop.frame()
        stats = ctx.stats_read(reset=True)
        ctx.stats_enable(False)
        torch.cuda.synchronize()

        reduce_device = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=reduce_device)
        work = torch.tensor([stats["rays"], stats["nodes"], stats["triangles"], stats["shaded"]], dtype=torch.float64, device=reduce_device)
        if multi:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_reduce(work, op=dist.ReduceOp.SUM)
        elapsed = float(t.item())
        rays_total = float(work[0].item())

        if rank == 0:
            wavefront = "wavefront.extend" in prof
            dominant = "wavefront.extend" if wavefront else "compute.megakernel"
            k = prof.get(dominant, dict(launches=0, total_ms=0.0))
            node_bytes, tri_bytes = (64, 64) if wavefront else (32, 48)
            algo_bytes = node_bytes * stats["nodes"] + tri_bytes * stats["triangles"]  # this rank's launches
            achieved = algo_bytes / (k["total_ms"] * 1e-3) / 1e9 if k["total_ms"] > 0 else 0.0
            line_workload = (f"{sc['kind']} (Bistro-Exterior-class procedural stand-in), {sc['triangles'].shape[0] * int(sc.get('copies', 1))} triangles, "
                             f"{W}x{H}, 1 spp per step, max_bounce {args.bounces}, min_bounce 1, dof off")
            ks = solo.get(dominant, dict(launches=0, total_ms=0.0))
            solo_bytes = algo_bytes * ks["launches"] / max(1, k["launches"])
            traffic, traffic_note, counters = pmc_traffic(dominant, line_workload)
            line = {
                "metric": "Mrays/s", "value": rays_total / elapsed / 1e6, "unit": "Mrays/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": line_workload,
                           "pipeline": "wavefront over flattened BVH" if wavefront else "megakernel over threaded BVH",
                           "partition": "samples" if world > 1 else "none", "frames_in_flight": args.frames_in_flight, "bvh_builder": args.bvh,
                           "instancing": "none" if args.instances in ("", "1x1") else (f"{args.instances} copies through a top-level tree" if args.two_level else f"{args.instances} copies written out"),
                           "rays_per_step_per_gpu": stats["rays"] / args.steps},
                "ms_per_frame": elapsed / args.steps * 1e3,
                "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                             "launches": k["launches"], "avg_launch_ms": k["total_ms"] / max(1, k["launches"]),
                             "algorithmic_bytes_per_launch": algo_bytes / max(1, k["launches"]),
                             "bytes_per_unit": {"node": node_bytes, "triangle": tri_bytes},
                             "nodes_per_ray": stats["nodes"] / max(1, stats["rays"]), "triangles_per_ray": stats["triangles"] / max(1, stats["rays"]),
                             "traffic_note": traffic_note,
                             # from the same counter passes: VALU issue utilisation, cache hit rates, share of wave cycles spent waiting
                             "counters": counters,
                             # the same kernel with the GPU to itself (one frame in flight): launches of overlapped frames share the CUs
                             "solo": {"avg_launch_ms": ks["total_ms"] / max(1, ks["launches"]),
                                      "achieved": solo_bytes / (ks["total_ms"] * 1e-3) / 1e9 if ks["total_ms"] > 0 else 0.0,
                                      "frac": (solo_bytes / (ks["total_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if ks["total_ms"] > 0 else 0.0}},
                "kernels_ms_per_step_solo": {n: v["total_ms"] / solo_frames for n, v in sorted(solo.items()) if v["launches"]},
                "setup": build_times,
            }
            if world == 1 and args.cpu_seconds > 0:
                line["cpu_baseline"] = cpu_baseline(args, sc, nodes, flat, meta, gpu, args.cpu_seconds)
                line["cpu_baseline"]["reference_on_lavapipe"] = lavapipe_probe()
            else:
                line["cpu_baseline"] = None
            os.write(result_fd, (json.dumps(line) + "\n").encode())
        ctx.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
