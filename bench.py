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
L2_PEAK_GBS = 34500.0           # aggregate L2 bandwidth of the 8 XCDs, measured
INFINITY_CACHE_GATHER_GBS = 8600.0   # uniformly random rows of a 38 MB table (served by the 256 MB Infinity Cache), measured chip-wide
CUS, SIMDS = 256, 1024
# VALU issue peak.  A gfx950 SIMD has two 16-lane vector pipes: measured with scripts/microbench/valu_issue.hip, a wave64
# instruction issues every 2.3 cycles when the stream is fp32 fma / mul / add, integer add, and / or / xor, right shifts or moves
# on vector registers (either pipe takes them), and every 4.2 cycles when it is v_cndmask, v_cmp, v_min / v_max, a conversion, a
# left shift, a packed fp32 form or anything with a scalar-register operand (one pipe only).  The peak is both pipes:
# 256 CUs x 4 SIMDs x 32 lanes per cycle at the 2.4 GHz peak engine clock, in lane-instructions per second (= the fp32 vector
# peak of 157 TFLOP/s counted as instructions).
VALU_PEAK_TLANE = CUS * 4 * 32 * 2.4e9 / 1e12
# VALU instructions of one 4-wide node step (four quantised box tests, three compare-exchanges, push) and of one triangle test on
# their common paths in the compiled traversal kernel (counted in the gfx950 ISA of csrc/ptr_kernels_wavefront.hip; of the node
# step's 125, 87 are of the one-pipe class)
VALU_PER_NODE, VALU_PER_TRIANGLE = 125, 66


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def load_or_generate(args, rank):
    """The scene of the line: the reference's own asset when its files are there AND are the objects its Git-LFS pointers name
    (path_tracer_amd/assets.py), through the glTF loader; the procedural stand-in otherwise.  sc["data"] says which."""
    from path_tracer_amd import assets, host, scenes
    path, report = assets.probe(getattr(args, "models_dir", None) or os.path.join(ROOT, "models"), args.scene)
    if path is None:
        log(rank, f"real assets: not used ({report.get('reason')}); synthetic scene")
        sc = scenes.generate(args.scene, args.triangles, texture_size=args.texture_size)
        sc["data"] = "synthetic"
        return sc
    sc = host.gltf_load(path, apply_transforms=True)
    folder = assets.SCENES[args.scene][0]
    # the reference's one camera (src/main.cpp:219-223)
    sc.update(kind=f"{folder} (the reference's asset, {report['files_verified']} files sha256-verified against its Git-LFS pointers)", data="real",
              camera=dict(pos=[13.0, 2.0, -3.0], target=[0.0, 0.0, 0.0], v_fov=90.0, aperture=0.1, focus=10.0))
    log(rank, f"real assets: {path} ({report['files_verified']} files verified)")
    return sc


def build_scene(args, rank):
    from path_tracer_amd import host, scenes
    t0 = time.time()
    sc = load_or_generate(args, rank)
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
    extra = {k: rec[k] for k in ("valu_one_pipe_share", "l2_hit_rate", "l1_hit_rate", "waves_waiting", "per_launch") if k in rec}
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
    ap.add_argument("--models-dir", default=None, help="where the reference's real scene files would be (default: models/ next to this file); used only when "
                                                       "every file matches the sha256 of the reference's Git-LFS pointer, else the procedural stand-in is rendered")
    ap.add_argument("--instances", default="1x1", help="NXxNZ translated copies of the scene on a grid (e.g. 6x6 of the default scene = 10^8 triangles)")
    ap.add_argument("--two-level", action="store_true", help="with --instances: keep ONE copy of the scene and its tree, and place it through a top-level tree")
    ap.add_argument("--pipeline", default="compute", choices=["compute", "compute.megakernel", "compute.wavefront"])
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--skip-parity", action="store_true", help="leave out the comparison of the timed region's image with the CPU oracle (a few seconds of host time)")
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
    counters = {k: v for k, v in counters.items() if k != "per_launch"}
    out = {"kernel": name, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": min(1.0, achieved / HBM_PEAK_GBS),
           "traffic": traffic, "launches": launches, "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": algo_bytes_total / max(1, launches),
           "algorithmic_bytes": note, "counters": counters}
    if achieved > HBM_PEAK_GBS:   # more than the memory can deliver: the bytes come out of the caches, the memory side moves `traffic`
        out["served_by_caches"] = True
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
    stand_in = "" if sc.get("data") == "real" else " (Bistro-Exterior-class procedural stand-in)"
    line_workload = (f"{sc['kind']}{stand_in}, {sc['triangles'].shape[0] * int(sc.get('copies', 1))} triangles, "
                     f"{W}x{H}, 1 spp per step, max_bounce {args.bounces}, min_bounce 1, dof off")
    ks = solo.get(dominant, dict(launches=0, total_ms=0.0))
    solo_bytes = algo_bytes * ks["launches"] / max(1, k["launches"]) if k["launches"] else algo_bytes
    traffic, traffic_note, counters = pmc_traffic(dominant, line_workload)
    if world > 1:   # the counter passes were taken on one GPU rendering whole frames: a rank's launches move other byte counts
        traffic, traffic_note, counters = None, "profiles/traffic.json holds one-GPU counter passes; not applied to a rank of several", {}
    avg_ms = k["total_ms"] / max(1, k["launches"])
    solo_ms = ks["total_ms"] / max(1, ks["launches"])
    # lanes that held a ray, per inner-node step of a traversal wave (counting variant of the same kernel)
    lane_util = stats["nodes"] / (64.0 * stats["wave_inner_steps"]) if stats.get("wave_inner_steps") else None
    partition = "none" if world == 1 else ("tiles" if tiles else "samples")
    launches = k["launches"] or ks["launches"] or 1
    solo_s = solo_ms * 1e-3
    algo_bytes_per_launch = algo_bytes / launches
    traffic_part = {k2: v for k2, v in counters.items() if k2 != "per_launch"}
    if wavefront:
        # The roofline that binds the traversal kernel on a scene the caches hold: instruction issue.  Algorithmic
        # lane-instructions (node steps and triangle tests only: no bookkeeping, every lane counted as useful) over the SOLO
        # launch time, against what the SIMDs can issue.
        lane_instr_per_launch = (VALU_PER_NODE * stats["nodes"] + VALU_PER_TRIANGLE * stats["triangles"]) / launches
        valu_achieved = lane_instr_per_launch / solo_s / 1e12 if solo_s > 0 else 0.0
        roofline = {"bound": "valu", "kernel": dominant, "achieved": valu_achieved, "peak": VALU_PEAK_TLANE, "unit": "Tlane-instr/s",
                    "frac": valu_achieved / VALU_PEAK_TLANE, "traffic": traffic,
                    "denominator": "solo launches (one dispatch in flight), HIP events on the stream the kernel runs on",
                    "instructions_per_unit": {"node": VALU_PER_NODE, "triangle": VALU_PER_TRIANGLE},
                    "peak_note": "two 16-lane vector pipes per SIMD at 2.4 GHz; only fma / mul / add / logic instructions on vector registers can use both (scripts/microbench/valu_issue.hip)"}
        pl = counters.get("per_launch")
        if pl and pl.get("GRBM_GUI_ACTIVE"):
            cycles = pl["GRBM_GUI_ACTIVE"] / 8.0                    # the counter sums the 8 XCDs
            issued = pl.get("SQ_INSTS_VALU", 0.0)
            from_counters = {"cycles_per_launch": cycles,
                             "issued_frac": issued * 64.0 / (SIMDS * 32.0 * cycles),
                             "lanes_active": pl["SQ_THREAD_CYCLES_VALU"] / (64.0 * pl["SQ_ACTIVE_INST_VALU"]) if pl.get("SQ_ACTIVE_INST_VALU") and pl.get("SQ_THREAD_CYCLES_VALU") else None,
                             "useful_over_issued": lane_instr_per_launch / (issued * 64.0) if issued else None,
                             "frac": lane_instr_per_launch / (SIMDS * 32.0 * cycles),
                             "one_pipe_share": pl.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / (SIMDS * cycles),
                             "note": "replayed from profiles/traffic.json (separate rocprofv3 --pmc passes over this same command, taken when the file was made): "
                                     "issued_frac = SQ_INSTS_VALU x 64 lanes over both pipes' lane-cycles; lanes_active = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); "
                                     "frac = this run's algorithmic lane-instructions over the counter passes' cycles; one_pipe_share = the share of cycles ONE pipe "
                                     "would be busy if it issued every vector instruction (what binds a stream of one-pipe-class instructions)"}
            roofline["from_counters"] = from_counters
            roofline["vector_memory"] = {"ta_busy": pl["TA_TA_BUSY_sum"] / (CUS * cycles) if pl.get("TA_TA_BUSY_sum") else None,
                                         "ta_stalled_by_l1_share": pl["TA_ADDR_STALLED_BY_TC_CYCLES_sum"] / pl["TA_TA_BUSY_sum"] if pl.get("TA_TA_BUSY_sum") and pl.get("TA_ADDR_STALLED_BY_TC_CYCLES_sum") else None,
                                         "l1_accesses_per_cycle_per_cu": pl["TCP_TOTAL_CACHE_ACCESSES_sum"] / (CUS * cycles) if pl.get("TCP_TOTAL_CACHE_ACCESSES_sum") else None,
                                         "load_instructions_per_launch": pl.get("SQ_INSTS_VMEM_RD"),
                                         "note": "the second resource in sight: a lane's 64-byte node is four 16-byte loads, each an L1 access of its own because the lanes of a wave sit at different nodes"}
        roofline["cache_side"] = {"achieved": algo_bytes_per_launch / solo_s / 1e9 if solo_s > 0 else 0.0, "unit": "GB/s",
                                  "algorithmic_bytes_per_launch": algo_bytes_per_launch,
                                  "bytes_per_unit": {"node": node_bytes, "triangle": tri_bytes},
                                  "note": "algorithmic bytes (64 B per node fetched + 64 B per triangle fetched) over the solo launch time: served by L1 / L2 / Infinity Cache, "
                                          "so it is read against their rates, not against HBM"}
        roofline["cache_side"]["vs_l2_peak"] = roofline["cache_side"]["achieved"] / L2_PEAK_GBS
        roofline["cache_side"]["vs_infinity_cache_peak"] = roofline["cache_side"]["achieved"] / INFINITY_CACHE_GATHER_GBS
        roofline["hbm"] = {"traffic": traffic, "achieved": traffic / solo_s / 1e9 if traffic and solo_s > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": traffic / solo_s / 1e9 / HBM_PEAK_GBS if traffic and solo_s > 0 else None,
                           "traffic_over_algorithmic": traffic / algo_bytes_per_launch if traffic and algo_bytes_per_launch else None,
                           "traffic_note": ("replayed from profiles/traffic.json (counter passes recorded when that file was made, not collected in this run); " + traffic_note) if traffic else traffic_note}
    else:
        roofline = {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": min(1.0, achieved / HBM_PEAK_GBS),
                    "traffic": traffic, "traffic_note": traffic_note, "algorithmic_bytes_per_launch": algo_bytes_per_launch,
                    "bytes_per_unit": {"node": node_bytes, "triangle": tri_bytes}}
    roofline.update({"launches": k["launches"], "avg_launch_ms": avg_ms, "solo_avg_launch_ms": solo_ms,
                     "nodes_per_ray": stats["nodes"] / max(1, stats["rays"]), "triangles_per_ray": stats["triangles"] / max(1, stats["rays"]),
                     "counts_note": "nodes and triangles are counted by the counting variant of the same kernel, which walks every ray with one lane to its closest hit; "
                                    "the timed variant finishes each wave's last ray with all 64 lanes (a few more nodes on those rays) and ends the rays of a "
                                    "path's last bounce at their first hit (fewer)",
                     "limiter": "instruction issue on the one-pipe class under divergence together with the rate of L1 accesses (vector_memory): the BVH is served by L2 / Infinity Cache "
                                "and the memory side moves a third of the algorithmic bytes, also on the 100 M-triangle scene",
                     "lane_utilisation": lane_util, "counters": traffic_part})
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
        # amortised: K steps rendered as batched dispatches with two dispatches in flight; one frame alone is single_frame_latency_ms
        "ms_per_frame": elapsed / args.steps * 1e3,
        "ms_per_frame_note": "throughput figure: the timed region's wall time / K, with several samples per dispatch and two dispatches in flight; "
                             "the latency of one sample rendered alone is single_frame_latency_ms, the reference's one-dispatch-per-sample protocol is one_sample_per_dispatch",
        "roofline": roofline,
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
        line["roofline"]["frame_level"] = {"algorithmic_bytes_per_step": total / args.steps, "achieved": total / elapsed / 1e9, "unit": "GB/s",
                                           "vs_l2_peak": total / elapsed / 1e9 / L2_PEAK_GBS,
                                           "note": "every kernel's algorithmic bytes over the step time; most of them are served by the caches (see roofline.hbm for what reaches the memory side)"}
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


def launch_ranks(n):
    """`python3 bench.py --gpus N` without a launcher's environment: this process starts the N ranks itself -- one copy of this
    command per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, as torch.distributed.run would -- and
    does so BEFORE anything here touches a GPU (a process that has initialised the GPU must not start or become another
    program on this pool; counting devices does not initialise it).  Rank 0's standard output, the one JSON line, is this
    process's; the other ranks' goes to standard error.  Returns the exit status: non-zero as soon as any rank fails."""
    import socket
    import subprocess
    for var in ("ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD", "HSA_TOOLS_LIB"):
        if os.environ.get(var):
            # a profiler's preloaded library has initialised the GPU in this process already: starting the ranks from it is
            # exactly what must not happen.  Profile one rank instead (the launcher's environment set by hand).
            print(f"[bench] --gpus {n} under a profiler ({var} is set): this process would start the ranks after the GPU was initialised. "
                  f"Profile one rank: RANK=r LOCAL_RANK=r WORLD_SIZE={n} MASTER_ADDR=127.0.0.1 MASTER_PORT=p <profiler> -- python3 bench.py --gpus {n} ...",
                  file=sys.stderr)
            return 2
    import torch
    if not os.environ.get("PTR_BENCH_DRY_RUN") and "PTR_BENCH_DEVICE" not in os.environ:
        found = torch.cuda.device_count()
        if found < n:
            print(f"[bench] --gpus {n}: this machine has {found} GPU(s).  (To rehearse the ranks on one GPU: PTR_BENCH_DEVICE=0 "
                  f"PTR_BENCH_BACKEND=gloo.)", file=sys.stderr)
            return 2
    with socket.socket() as sock:                       # a free port for the rendezvous
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    status = 0
    live = list(procs)
    while live:
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0 and status == 0:
                status = rc if rc > 0 else 1
                for q in live:                           # a rank that failed leaves the others waiting in a collective
                    q.terminate()
        time.sleep(0.05)
    return status


def dry_run(args, world, rank):
    """PTR_BENCH_DRY_RUN=1: everything of a multi-rank run that needs no GPU -- the launcher, the rendezvous, the barriers and the
    max-over-ranks reduction -- with nothing rendered.  The line says so (`dry_run`, value null); it exists for the CPU test of the
    launcher and is never a measurement."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    dist.init_process_group(os.environ.get("PTR_BENCH_BACKEND", "gloo"), rank=rank, world_size=world)
    dist.barrier()
    t0 = time.perf_counter()
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    seen = dist.get_world_size()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        return {"metric": "Mrays/s", "value": None, "unit": "Mrays/s", "n_gpus": world, "ranks_seen": seen, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": None, "higher_is_better": True, "scaling": "strong" if args.partition == "tiles" else "weak", "vs_baseline": None,
                "dtype": "f32", "data": "none", "dry_run": True, "barrier_ms": float(t[0].item()) * 1e3,
                "config": {"workload": "nothing rendered: launcher and rendezvous only (PTR_BENCH_DRY_RUN)", "partition": args.partition}}
    return None


def main():
    args = make_parser().parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    # Rank 0 must print ONE line on stdout.  Libraries underneath write there too (RCCL prints a version banner when the
    # first communicator is created): everything written to fd 1 from here on goes to stderr, the result line goes to the
    # saved descriptor.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(rank, f"note: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE")
    if os.environ.get("PTR_BENCH_DRY_RUN"):
        line = dry_run(args, world, rank)
        if line is not None:
            os.write(result_fd, (json.dumps(line) + "\n").encode())
        return

    import torch
    import torch.distributed as dist
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
            flat = gpu_flat.read(np.uint8) if (args.cpu_seconds > 0 or not args.skip_parity) else flat
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

        exchange_events = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        exchange_ms = [None]

        def timed(steps, with_exchange):
            """`steps` frames from sample 1 on cleared images, barrier + device sync on both sides; returns (seconds, final image).
            exchange_ms[0] = device time of the partition's exchange (pack + collective + unpack), HIP events on torch's stream, which
            is the stream the exchange runs on (with --stream own it misses the library's pack / unpack kernels)."""
            loop.reset()
            img_a.zero_(); img_b.zero_()
            barrier()
            t0 = time.perf_counter()
            loop.frames(steps)
            ctx.flush()    # dispatches held back under "dispatch.coalesce" are enqueued now (nothing to do otherwise)
            if with_exchange:
                exchange_events[0].record()
                final = combine()
                exchange_events[1].record()
            else:
                final = by_handle[loop.latest().handle]
            torch.cuda.synchronize()
            if multi:
                dist.barrier()
            dt = time.perf_counter() - t0
            exchange_ms[0] = exchange_events[0].elapsed_time(exchange_events[1]) if with_exchange else None
            return dt, final

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
        timed_exchange_ms = exchange_ms[0]
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
        # one sample rendered alone: one dispatch of one sample, nothing else in flight, host waits for it (the latency a caller
        # sees who needs every frame before it issues the next; not a throughput figure)
        single_frame_latency_ms = None
        if not os.environ.get("PTR_BENCH_SKIP_SINGLE"):   # (skipped under counter passes, like the one-sample leg below)
            keep_spd = loop.samples_per_dispatch
            loop.samples_per_dispatch = 1
            loop.reset()
            loop.frames(2); torch.cuda.synchronize()
            lat = []
            for _ in range(min(args.steps, 10)):
                t_l = time.perf_counter()
                loop.frames(1)
                torch.cuda.synchronize()
                lat.append(time.perf_counter() - t_l)
            single_frame_latency_ms = sorted(lat)[len(lat) // 2] * 1e3
            loop.samples_per_dispatch = keep_spd
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
        t = torch.tensor([elapsed, t1_elapsed or 0.0, timed_exchange_ms or 0.0], dtype=torch.float64, device=reduce_device)
        work = torch.tensor([stats["rays"], stats["nodes"], stats["triangles"], stats["shaded"]], dtype=torch.float64, device=reduce_device)
        if multi:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_reduce(work, op=dist.ReduceOp.SUM)
        elapsed, t1_elapsed, timed_exchange_ms = float(t[0].item()), float(t[1].item()), float(t[2].item())
        rays_total = float(work[0].item())

        if rank == 0:
            line = result_line(args, sc, W, H, world, tiles, elapsed, t1_elapsed, rays_total, stats, prof, solo, solo_frames, build_times)
            line["config"]["samples_per_dispatch"] = loop.samples_per_dispatch
            line["config"]["dispatches_per_gpu"] = -(-args.steps // loop.samples_per_dispatch)
            line["config"]["warmup_steps_run"] = warmup_steps
            line["ranks_seen"] = dist.get_world_size() if multi else 1
            line["exchange_ms"] = timed_exchange_ms if multi else None
            line["single_frame_latency_ms"] = single_frame_latency_ms
            line["data"] = sc.get("data", "synthetic")
            if single_ms is not None:
                line["one_sample_per_dispatch"] = {"ms_per_step": single_ms, "value": rays_total / (single_ms * 1e-3 * args.steps) / 1e6, "unit": "Mrays/s",
                                                    "note": "the same K steps as K dispatches of one sample each (the reference's frame protocol), same partition and exchange, outside the timed region",
                                                    "coalesced": {"ms_per_step": coalesced_ms, "value": rays_total / (coalesced_ms * 1e-3 * args.steps) / 1e6, "unit": "Mrays/s",
                                                                  "dispatch.coalesce": loop.samples_per_dispatch, "image_equals_timed_region": coalesced_equal,
                                                                  "note": "the same K single-sample dispatches with \"dispatch.coalesce\" set: the library holds them back and renders them as batched dispatches"}}
            if timed_image is not None and args.pipeline != "compute.megakernel" and flat is not None and not args.skip_parity:
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
