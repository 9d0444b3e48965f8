#!/usr/bin/env python3
"""Headline benchmark of the render hot path: Mrays/s and ms/frame, 1920x1080, 1 sample per step, on a
Bistro-Exterior-class synthetic scene (the reference's real assets are Git-LFS pointers, unavailable offline).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one full-resolution 1-spp pass of the "compute" dispatch (generate + per-bounce extend/shade +
accumulate), inputs resident in HBM.  With N GPUs the scene is replicated and the work is partitioned:

  --partition tiles (default)   every frame is split: rank r renders the 8-row stripes s with s % N == r of each of the K
                                frames, one RCCL all-gather of the owned stripes closes the timed region.  Total work is
                                fixed, so the scaling is strong; the merged frame is bit-identical to the one-GPU frame.
  --partition samples           rank r renders whole frames for the global samples r+1, r+1+N, ...; one RCCL all-reduce of
                                the linearized images closes the timed region.  Per-GPU work is fixed: weak scaling.

Rank 0 prints one JSON line.
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
# VALU issue peak: 256 CUs x 4 SIMDs x 16 lanes per cycle at the 2.4 GHz peak engine clock, in lane-instructions per second
VALU_PEAK_TLANE = 256 * 4 * 16 * 2.4e9 / 1e12
# VALU instructions of one 4-wide node step (four quantised box tests, three compare-exchanges, push) and of one triangle test on their common
# paths in the compiled traversal kernel (counted in the gfx950 ISA of csrc/ptr_kernels_wavefront.hip)
VALU_PER_NODE, VALU_PER_TRIANGLE = 113, 67


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
        flat = None if getattr(args, "bvh", "host") in ("gpu", "auto") else host.bvh_flatten(sc["triangles"], None, threads=min(16, os.cpu_count() or 1))
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
    flat = None if getattr(args, "bvh", "host") == "auto" else host.bvh_flatten(sc["triangles"], ranks, threads=min(16, os.cpu_count() or 1))
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
        if time.time() - t1 >= max(1.0, budget_s * 0.2):
            break
    single_dt = time.time() - t1
    result = dict(value=rays / dt / 1e6, unit="Mrays/s", cores=threads, kind="port", single_thread_value=single_rays / single_dt / 1e6,
                  sample=f"{rows_done} rows of {H}-row frames ({rows_done // band} bands of {band} rows spread over the frame, sample_index 1..{sample_index - 1}), "
                         f"{rays} rays in {dt:.1f}s, " + ("flattened BVH" if use_flat else "reference-order threaded BVH"),
                  tree="flattened 4-wide SAH layout" if use_flat else "the reference builder's threaded tree, walked like shaders/include/ray.h:165-193")
    if not use_flat and flat is not None:
        # like for like with the GPU figure: the same port walking the SAME flattened layout the kernels walk (the reference's
        # own tree costs ~40x the nodes per ray, so most of the GPU / CPU ratio above is the better tree, not the hardware)
        flat_rays, t2, passes = 0, time.time(), 0
        while time.time() - t2 < max(2.0, budget_s * 0.4) and passes < 64:
            o.set_meta(sample_index=passes + 1)
            for y in order:
                flat_rays += o.render(acc, out, rows=(y, y + band), threads=threads, use_flat=True)["rays"]
                if time.time() - t2 >= max(2.0, budget_s * 0.4):
                    break
            acc, out = out, acc
            passes += 1
        flat_dt = time.time() - t2
        result["same_tree_as_gpu"] = dict(value=flat_rays / flat_dt / 1e6, unit="Mrays/s", cores=threads,
                                          sample=f"{flat_rays} rays in {flat_dt:.1f}s on the same bands, sample_index 1..{passes}, flattened 4-wide SAH layout")
    return result


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


def pmc_traffic(kernel, workload, samples_per_dispatch=None):
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
    if samples_per_dispatch is None:
        samples_per_dispatch = pmc_traffic.samples_per_dispatch
    if rec is None or t.get("workload") != workload:
        return None, "profiles/traffic.json holds no counters for this kernel and workload", {}
    extra = {k: rec[k] for k in ("valu_busy", "l2_hit_rate", "l1_hit_rate", "waves_waiting") if k in rec}
    recorded = t.get("samples_per_dispatch", 1)
    if recorded == samples_per_dispatch:
        return rec["bytes_per_launch"], t.get("note", ""), extra
    if recorded > 1 and samples_per_dispatch > 1:
        # launches of another size: the bytes of a launch grow with the samples in it (the counters were taken per launch)
        scaled = rec["bytes_per_launch"] * samples_per_dispatch / recorded
        return scaled, (f"scaled from the counter passes' launches of {recorded} samples to this run's {samples_per_dispatch} "
                        f"(bytes per sample taken as equal); " + t.get("note", "")), extra
    return None, "profiles/traffic.json holds counters for another dispatch size (one sample per dispatch against several)", {}


pmc_traffic.samples_per_dispatch = 1   # set by main(): launches of another size move other byte counts


def make_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
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
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--bvh", default="host", choices=["host", "gpu", "auto"],
                    help="host: binned-SAH flattener (default); gpu: ptr_bvh_build_flat; auto: only the reference's buffers are uploaded and "
                         "\"bvh.auto_flatten\" lets the first dispatch build and bind the layout")
    ap.add_argument("--stream", default="torch", choices=["torch", "own"], help="the context's stream: torch's current stream or its own")
    ap.add_argument("--frames-in-flight", type=int, default=2, help="consecutive frames overlapped on internal streams (1 = none)")
    ap.add_argument("--partition", default="tiles", choices=["tiles", "samples"], help="how N > 1 GPUs share the work (see the module text)")
    ap.add_argument("--stripe-rows", type=int, default=8, help="tile partition: image rows per stripe (a multiple of 8)")
    ap.add_argument("--samples-per-dispatch", type=int, default=16,
                    help="upper bound of the samples (= steps) one dispatch renders (\"dispatch.samples\"), times the number of GPUs in the tile "
                         "partition (64 at most): the K steps are dealt evenly to as few dispatches as that allows; the image equals K "
                         "single-sample dispatches bit for bit.  1 = the reference's one dispatch per sample")
    return ap


def kernel_roofline(name, algo_bytes_total, prof_rec, workload, note):
    """One kernel's line: algorithmic bytes of all its launches / their summed HIP-event time, against the HBM peak, with the
    counter passes' memory-side bytes beside it."""
    launches, ms = prof_rec["launches"], prof_rec["total_ms"]
    achieved = algo_bytes_total / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    traffic, _, counters = pmc_traffic(name, workload)
    avg_ms = ms / max(1, launches)
    out = {"kernel": name, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "traffic": traffic, "launches": launches, "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": algo_bytes_total / max(1, launches),
           "algorithmic_bytes": note, "counters": counters}
    if traffic and avg_ms > 0:
        out["traffic_over_algorithmic"] = traffic / (algo_bytes_total / max(1, launches))
        out["hbm_frac_measured"] = traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    return out


def result_line(args, sc, W, H, world, tiles, elapsed, t1_elapsed, rays_total, stats, prof, solo, solo_frames, build_times):
    wavefront = "wavefront.extend" in prof or "wavefront.extend" in solo
    dominant = "wavefront.extend" if wavefront else "compute.megakernel"
    k = prof.get(dominant, dict(launches=0, total_ms=0.0))
    node_bytes, tri_bytes = (64, 64) if wavefront else (32, 48)
    algo_bytes = node_bytes * stats["nodes"] + tri_bytes * stats["triangles"]  # this rank's launches over all timed steps
    achieved = algo_bytes / (k["total_ms"] * 1e-3) / 1e9 if k["total_ms"] > 0 else 0.0
    line_workload = (f"{sc['kind']} (Bistro-Exterior-class procedural stand-in), {sc['triangles'].shape[0] * int(sc.get('copies', 1))} triangles, "
                     f"{W}x{H}, 1 spp per step, max_bounce {args.bounces}, min_bounce 1, dof off")
    ks = solo.get(dominant, dict(launches=0, total_ms=0.0))
    solo_bytes = algo_bytes * ks["launches"] / max(1, k["launches"]) if k["launches"] else algo_bytes
    traffic, traffic_note, counters = pmc_traffic(dominant, line_workload)
    avg_ms = k["total_ms"] / max(1, k["launches"])
    solo_ms = ks["total_ms"] / max(1, ks["launches"])
    # lanes that held a ray, per inner-node step of a traversal wave (counting variant of the same kernel)
    lane_util = stats["nodes"] / (64.0 * stats["wave_inner_steps"]) if stats.get("wave_inner_steps") else None
    partition = "none" if world == 1 else ("tiles" if tiles else "samples")
    line = {
        "metric": "Mrays/s", "value": rays_total / elapsed / 1e6, "unit": "Mrays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if (tiles or world == 1) else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": line_workload,
                   "pipeline": "wavefront over flattened BVH" if wavefront else "megakernel over threaded BVH",
                   "partition": partition, "exchange": {"none": "none", "tiles": "one all-gather of the owned stripes inside the timed region",
                                                        "samples": "linearize + one all-reduce + resolve inside the timed region"}[partition],
                   "frames_in_flight": args.frames_in_flight, "bvh_builder": args.bvh,
                   "instancing": "none" if args.instances in ("", "1x1") else (f"{args.instances} copies through a top-level tree" if args.two_level else f"{args.instances} copies written out"),
                   "rays_per_step_per_gpu": stats["rays"] / args.steps},
        "ms_per_frame": elapsed / args.steps * 1e3,
        "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "launches": k["launches"], "avg_launch_ms": avg_ms,
                     "algorithmic_bytes_per_launch": algo_bytes / max(1, k["launches"]),
                     "bytes_per_unit": {"node": node_bytes, "triangle": tri_bytes},
                     "nodes_per_ray": stats["nodes"] / max(1, stats["rays"]), "triangles_per_ray": stats["triangles"] / max(1, stats["rays"]),
                     "counts_note": "nodes and triangles are counted by the counting variant of the same kernel, which walks every ray with one lane; "
                                    "the timed variant finishes each wave's last ray with all 64 lanes and can touch a few more nodes on those rays",
                     "traffic_note": traffic_note,
                     # What binds: `frac` prices the ALGORITHMIC bytes against HBM as the measurement contract asks, but this scene's
                     # BVH (40 MB of nodes + 179 MB of triangles) is served by L2 + the 256 MB Infinity Cache: the bytes that reach the
                     # memory side are `traffic` (hbm_frac_measured).  The kernel is bound by VALU issue under divergence
                     # (valu_roofline below, counters, lane_utilisation), not by HBM bandwidth.
                     "limiter": "VALU issue under divergence (valu_roofline, counters.valu_busy, lane_utilisation); the BVH is served by L2 / Infinity Cache and the memory side moves a third of the algorithmic bytes, also on the 100 M-triangle scene",
                     "hbm_frac_measured": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic and avg_ms > 0 else None,
                     "traffic_over_algorithmic": (traffic / (algo_bytes / max(1, k["launches"]))) if traffic and k["launches"] else None,
                     "lane_utilisation": lane_util,
                     "counters": counters,
                     # launches of overlapped frames share the CUs, so their event durations sum to more than the frame; the same
                     # kernel with the GPU to itself (one frame in flight) is the clean per-launch denominator
                     "solo": {"avg_launch_ms": solo_ms,
                              "achieved": solo_bytes / (ks["total_ms"] * 1e-3) / 1e9 if ks["total_ms"] > 0 else 0.0,
                              "frac": (solo_bytes / (ks["total_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if ks["total_ms"] > 0 else 0.0},
                     # the whole step against the clock: every kernel's algorithmic bytes / ms_per_step
                     "frame_level": None},
        # The roofline that does bind the traversal kernel on a cache-resident scene: VALU issue.  Algorithmic lane-instructions
        # (node steps and triangle tests only: no bookkeeping, every lane counted as useful) over the solo launch time, against
        # what the SIMDs can issue.  The gap to 1 is lanes without work in a step (lane_utilisation, leaf / inner divergence),
        # cycles without issue (counters.valu_busy) and the instructions around the two step bodies.
        "valu_roofline": ({"kernel": dominant, "bound": "valu", "unit": "Tlane-instr/s", "peak": VALU_PEAK_TLANE,
                           "achieved": (VALU_PER_NODE * stats["nodes"] + VALU_PER_TRIANGLE * stats["triangles"]) * (ks["launches"] / max(1, k["launches"]) if k["launches"] else 1.0)
                                       / (ks["total_ms"] * 1e-3) / 1e12 if ks["total_ms"] > 0 else 0.0,
                           "instructions_per_unit": {"node": VALU_PER_NODE, "triangle": VALU_PER_TRIANGLE},
                           "denominator": "solo launches (one dispatch in flight)"} if wavefront else None),
        "kernels_ms_per_step_solo": {n: v["total_ms"] / solo_frames for n, v in sorted(solo.items()) if v["launches"]},
        "setup": build_times,
    }
    if wavefront:
        pixels = W * H * args.steps // (world if tiles else 1)
        paths = stats["rays"]                        # one path record per ray traced
        survivors = max(0, stats["rays"] - pixels)   # rays of bounce >= 1 are the survivors some shade launch wrote
        # (primary rays carry no throughput record: 48 B instead of 64 B read for one path per pixel and sample)
        shade_bytes = 64 * paths - 16 * pixels + 48 * survivors + (80 + 48 + 32) * stats["shaded"] + 16 * pixels
        dispatches = -(-args.steps // max(1, pmc_traffic.samples_per_dispatch))
        acc_bytes = 16 * pixels + 32 * (pixels // args.steps) * dispatches   # every sample is read; the images once per dispatch
        gen_bytes = 32 * pixels
        others = []
        if solo.get("wavefront.shade", {}).get("launches"):
            others.append(kernel_roofline("wavefront.shade", shade_bytes, solo["wavefront.shade"], line_workload,
                                          "per path 64 B (origin, direction, throughput, hit; 48 B for a primary ray, whose throughput is 1 and not stored) + per survivor 48 B written + per shaded hit 80 B record + 48 B material + 2 x 4 texels + 16 B sample per finished path"))
        if solo.get("wavefront.accumulate", {}).get("launches"):
            others.append(kernel_roofline("wavefront.accumulate", acc_bytes, solo["wavefront.accumulate"], line_workload,
                                          "per sample 16 B sample read; per dispatch and pixel 16 B accumulation read + 16 B output write"))
        if solo.get("wavefront.generate", {}).get("launches"):
            others.append(kernel_roofline("wavefront.generate", gen_bytes, solo["wavefront.generate"], line_workload, "per pixel 32 B of queue records written (origin | pixel, direction | seed)"))
        line["roofline"]["other_kernels_solo"] = others
        total = algo_bytes + shade_bytes + acc_bytes + gen_bytes
        line["roofline"]["frame_level"] = {"algorithmic_bytes_per_step": total / args.steps, "achieved": total / elapsed / 1e9,
                                           "frac": total / elapsed / 1e9 / HBM_PEAK_GBS}
    if line.get("valu_roofline"):
        line["valu_roofline"]["frac"] = line["valu_roofline"]["achieved"] / VALU_PEAK_TLANE
    if world > 1 and tiles and t1_elapsed:
        line["strong_scaling"] = {"t1_ms_per_step": t1_elapsed / args.steps * 1e3, "tn_ms_per_step": elapsed / args.steps * 1e3,
                                  "efficiency": t1_elapsed / (world * elapsed),
                                  "note": "T1 = the same frames rendered whole by one GPU of this job (max over ranks), outside the timed region"}
    return line


def parity_check(sc, nodes, flat, meta, gpu_scene, image, samples, rows_per_band=2, bands=4):
    """Outside the timed region: the frame the timed region ended on (sample `samples` of its accumulation chain) against
    the CPU oracle on a few row bands, bit for bit.  The oracle walks the same flattened layout the kernels walk."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    pyoracle.build()
    H, W = image.shape[0], image.shape[1]
    arrays = dict(sc, meta=meta, nodes=nodes, flat=flat, materials=gpu_scene.materials_host, textures=gpu_scene.texture_table)
    o = pyoracle.OracleScene(arrays)
    starts = [int(round(i * (H - rows_per_band) / max(1, bands - 1))) for i in range(bands)]
    acc = np.zeros((H, W, 4), np.float32)
    out = np.zeros_like(acc)
    t0 = time.time()
    for s in range(1, samples + 1):
        o.set_meta(sample_index=s)
        for y in starts:
            o.render(acc, out, rows=(y, y + rows_per_band), threads=min(os.cpu_count() or 1, 16), use_flat=True)
        acc, out = out, acc
    differing = 0
    for y in starts:
        differing += int((image[y:y + rows_per_band] != acc[y:y + rows_per_band]).any(axis=-1).sum())
    return dict(rows=[[y, y + rows_per_band] for y in starts], samples=samples, pixels=len(starts) * rows_per_band * W,
                differing_pixels=differing, equal=differing == 0, oracle_seconds=time.time() - t0,
                against="CPU oracle (oracle/oracle.c) walking the same flattened layout, same seeds")


def main():
    args = make_parser().parse_args()

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
    # PTR_BENCH_FORCE_DIST=1: run the exchange path (process group, barriers, the collective) with ONE rank,
    # to rehearse the RCCL leg on a one-GPU box
    multi = world > 1 or bool(os.environ.get("PTR_BENCH_FORCE_DIST"))
    tiles = args.partition == "tiles"
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
        if args.bvh == "auto":
            ctx.set_option("bvh.auto_flatten", 1)
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
        # K steps in ceil(K / S) dispatches of (nearly) equal size
        def per_dispatch(steps):
            # a rank of N renders 1/N of every frame: N times the samples per dispatch keep its launches as full -- but never
            # fewer than two dispatches, which is what lets one dispatch's shade and drain phases run beside the other's
            # traversal (one dispatch of 20 samples on a rank of 8: 3.26 ms, two of 10: 3.02 ms)
            limit = 1 if args.pipeline == "compute.megakernel" else max(1, min(64, args.samples_per_dispatch * (world if tiles else 1)))
            dispatches = max(min(2, steps), -(-steps // limit))
            return max(1, -(-steps // dispatches))
        loop = hip.FrameLoop(ctx, gpu, W, H, args.pipeline, textures=(tex_a, tex_b), samples_per_dispatch=per_dispatch(args.steps))
        pmc_traffic.samples_per_dispatch = loop.samples_per_dispatch

        def set_partition(on):
            if world > 1 and tiles:
                ctx.set_partition(rank if on else 0, world if on else 1, args.stripe_rows)
            elif world > 1:
                ctx.set_sample_partition(rank if on else 0, world if on else 1)
        set_partition(True)
        ctx.set_option("frames.in_flight", args.frames_in_flight)

        exchange = ctx.create_compute_renderpass()
        # the tile exchange needs the context on torch's stream (pack, all-gather and unpack are then one stream's work)
        tile_exchange = ptdist.TileExchange(ctx, W, H, rank, world, args.stripe_rows) if (multi and tiles and args.stream == "torch") else None

        def order_streams():
            """With the context on its own stream the collectives (torch's current stream) and the kernels around them are
            on different streams: the host waits in between.  With --stream torch everything is on one stream already."""
            if args.stream == "own":
                ctx.sync()
                torch.cuda.synchronize()

        def combine():
            """Closes the timed region with the exchange of the partition; returns the final image tensor.
            tiles: one all-gather of the owned stripes.  samples: linearize -> all-reduce -> resolve."""
            latest, scratch = loop.latest(), loop.output
            if tiles and tile_exchange is not None:
                tile_exchange.merge(latest)
                return by_handle[latest.handle]
            if tiles:
                order_streams()
                return ptdist.gather_tiles(by_handle[latest.handle], rank, world, args.stripe_rows)
            exchange.set_pipeline("linearize")
            exchange.set_ouput_texture(scratch)
            exchange.set_constant(60, latest)
            exchange.set_dispatch_size(1, 1, 1)
            exchange.execute()
            order_streams()
            ptdist.allreduce_linear(by_handle[scratch.handle])
            order_streams()
            exchange.set_pipeline("resolve")
            exchange.set_ouput_texture(latest)
            exchange.set_constant(60, scratch)
            exchange.execute()
            return by_handle[latest.handle]

        def barrier():
            if multi:
                dist.barrier()
            torch.cuda.synchronize()

        def timed(steps, with_exchange):
            """`steps` frames from sample 1 on cleared images, barrier + device sync on both sides; returns (seconds, final image)."""
            loop.reset()
            img_a.zero_(); img_b.zero_()
            barrier()
            t0 = time.perf_counter()
            loop.frames(steps)
            ctx.flush()    # dispatches held back under "dispatch.coalesce" are enqueued now (nothing to do otherwise)
            final = combine() if with_exchange else by_handle[loop.latest().handle]
            torch.cuda.synchronize()
            if multi:
                dist.barrier()
            return time.perf_counter() - t0, final

        # Untimed warm-up: the W steps asked for, and at least one dispatch of the timed size on each frame slot -- a slot's
        # scratch is sized by the largest dispatch it has seen, and growing it (hipFree + hipMalloc) belongs to set-up
        warmup_steps = max(args.warmup, args.frames_in_flight * loop.samples_per_dispatch)
        loop.frames(warmup_steps)
        if multi:
            combine()
        barrier()
        # live kernel timing inside the timed region: HIP events around the dominant kernel, on the stream it runs on
        ctx.profile_read(reset=True)
        ctx.set_option("profile.extend_only", 1)
        ctx.profile_enable(not os.environ.get("PTR_BENCH_NO_LIVE_PROFILE"))
        elapsed, final = timed(args.steps, multi)
        ctx.profile_enable(False)
        prof = ctx.profile_read(reset=True)
        ctx.set_option("profile.extend_only", 0)
        timed_image = final.cpu().numpy() if (rank == 0 and (tiles or world == 1)) else None   # (the legs below render into the same two images)

        # per-kernel split of a frame that has the GPU to itself (one frame in flight), outside the timed region -- behind the
        # host copy above and before the other legs: the same launches measure up to 7 % slower directly behind a leg that kept
        # the GPU fully busy (0.87 against 0.815 ms per sample for the traversal kernel), so the leg's place is fixed
        loop.reset()
        ctx.set_option("frames.in_flight", 1)
        ctx.profile_enable(True)
        solo_frames = args.steps              # the same sample indices as the timed frames: a frame's cost depends on its sample index
        loop.frames(solo_frames)
        torch.cuda.synchronize()
        ctx.profile_enable(False)
        solo = ctx.profile_read(reset=True)
        ctx.set_option("frames.in_flight", args.frames_in_flight)

        # what a caller pays on top when it reads every finished frame back over PCIe (33 MB at 1080p)
        t_rb = time.perf_counter()
        for _ in range(3):
            loop.latest().read()
        build_times["frame_readback_ms"] = (time.perf_counter() - t_rb) / 3 * 1e3
        t_rb = time.perf_counter()
        for _ in range(3):
            loop.latest().read_unorm8()       # ... and when it shows them: 8 bits per channel, converted on the device
        build_times["frame_readback_unorm8_ms"] = (time.perf_counter() - t_rb) / 3 * 1e3

        # the same K steps the way the reference's frame loop issues them: one dispatch per sample
        single_ms = None
        if loop.samples_per_dispatch > 1 and not os.environ.get("PTR_BENCH_SKIP_SINGLE"):   # (skipped under counter passes: every launch of a run should be of one size)
            keep = loop.samples_per_dispatch
            loop.samples_per_dispatch = 1
            single_elapsed, _ = timed(args.steps, multi)
            # ... and the same K single-sample dispatches with the library holding them back and rendering `keep` at a time
            ctx.set_option("dispatch.coalesce", keep)
            coalesced_elapsed, coalesced_final = timed(args.steps, multi)
            ctx.set_option("dispatch.coalesce", 1)
            coalesced_equal = bool(np.array_equal(coalesced_final.cpu().numpy(), timed_image)) if timed_image is not None else None
            loop.samples_per_dispatch = keep
            t_single = torch.tensor([single_elapsed, coalesced_elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            if multi:
                dist.all_reduce(t_single, op=dist.ReduceOp.MAX)
            single_ms = float(t_single[0].item()) / args.steps * 1e3
            coalesced_ms = float(t_single[1].item()) / args.steps * 1e3

        # strong scaling needs the one-GPU time of the SAME total work on the same hardware: every rank renders the whole
        # frames alone (no partition, no exchange), outside the timed region
        t1_elapsed = None
        if world > 1 and tiles:
            set_partition(False)
            t1_elapsed, _ = timed(args.steps, False)
            set_partition(True)

        # exact work of the timed frames: re-run the same sample indices with the counting kernel variants
        loop.reset()
        ctx.stats_read(reset=True)
        ctx.stats_enable(True)
        loop.frames(args.steps)
        stats = ctx.stats_read(reset=True)
        ctx.stats_enable(False)
        torch.cuda.synchronize()

        reduce_device = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([elapsed, t1_elapsed or 0.0], dtype=torch.float64, device=reduce_device)
        work = torch.tensor([stats["rays"], stats["nodes"], stats["triangles"], stats["shaded"]], dtype=torch.float64, device=reduce_device)
        if multi:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_reduce(work, op=dist.ReduceOp.SUM)
        elapsed, t1_elapsed = float(t[0].item()), float(t[1].item())
        rays_total = float(work[0].item())

        if rank == 0:
            line = result_line(args, sc, W, H, world, tiles, elapsed, t1_elapsed, rays_total, stats, prof, solo, solo_frames, build_times)
            line["config"]["samples_per_dispatch"] = loop.samples_per_dispatch
            line["config"]["dispatches_per_gpu"] = -(-args.steps // loop.samples_per_dispatch)
            line["config"]["warmup_steps_run"] = warmup_steps
            if single_ms is not None:
                line["one_sample_per_dispatch"] = {"ms_per_step": single_ms, "value": rays_total / (single_ms * 1e-3 * args.steps) / 1e6, "unit": "Mrays/s",
                                                    "note": "the same K steps as K dispatches of one sample each (the reference's frame protocol), same partition and exchange, outside the timed region",
                                                    "coalesced": {"ms_per_step": coalesced_ms, "value": rays_total / (coalesced_ms * 1e-3 * args.steps) / 1e6, "unit": "Mrays/s",
                                                                  "dispatch.coalesce": loop.samples_per_dispatch, "image_equals_timed_region": coalesced_equal,
                                                                  "note": "the same K single-sample dispatches with \"dispatch.coalesce\" set: the library holds them back and renders them as batched dispatches"}}
            if timed_image is not None and args.pipeline != "compute.megakernel" and flat is not None and args.cpu_seconds > 0:
                line["parity_check"] = parity_check(sc, nodes, flat, meta, gpu, timed_image, args.steps)
                log(rank, "parity_check:", line["parity_check"])
            else:
                line["parity_check"] = None
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
