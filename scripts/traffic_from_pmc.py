#!/usr/bin/env python3
"""Turns the FETCH_SIZE / WRITE_SIZE passes of scripts/pmc_passes.sh (run on bench.py) into profiles/traffic.json, the
file bench.py reads for roofline.traffic.
  usage: traffic_from_pmc.py <pmc dir> <bench json of the same command> <out json>
FETCH_SIZE is reported in KiB and, on gfx950, tallies a 128-byte memory-side read request as 64 bytes: it is doubled.
That correction is calibrated for wide streaming reads; the traversal kernel reads 64-byte records at scattered
addresses, for which the doubled figure is an upper bound.  WRITE_SIZE (KiB) is taken as is."""
import csv
import glob
import json
import os
import sys

NAMES = {"ptr_wf_extend<false": "wavefront.extend", "ptr_wf_shade<false": "wavefront.shade", "ptr_wf_generate": "wavefront.generate",
         "ptr_wf_accumulate": "wavefront.accumulate", "ptr_compute_megakernel<false": "compute.megakernel"}


RAW = {"sq1": ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VMEM_RD", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"),
       "sq2": ("SQ_THREAD_CYCLES_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_BRANCH"),
       "tcp": ("TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TA_TA_BUSY_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "GRBM_GUI_ACTIVE"),
       "tcc": ("TCC_HIT_sum", "TCC_MISS_sum")}


def mean_per_kernel(directory, counter):
    acc = {}
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter:
                    continue
                for prefix, name in NAMES.items():
                    if prefix in r["Kernel_Name"]:
                        acc.setdefault(name, []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    pmc_dir, bench_json, out = sys.argv[1:4]
    fetch, n = mean_per_kernel(os.path.join(pmc_dir, "fetch"), "FETCH_SIZE")
    write, _ = mean_per_kernel(os.path.join(pmc_dir, "write"), "WRITE_SIZE")
    with open(bench_json) as f:
        bench = json.loads(f.read().strip().splitlines()[-1])
    def counter(pass_name, counter_name):
        return mean_per_kernel(os.path.join(pmc_dir, pass_name), counter_name)[0]
    valu, busy = counter("sq1", "SQ_INSTS_VALU"), counter("tcp", "GRBM_GUI_ACTIVE")
    wait, wave_cycles = counter("sq1", "SQ_WAIT_ANY"), counter("sq1", "SQ_WAVE_CYCLES")
    hit, miss = counter("tcc", "TCC_HIT_sum"), counter("tcc", "TCC_MISS_sum")
    l1_req, l1_miss = counter("tcp", "TCP_TOTAL_CACHE_ACCESSES_sum"), counter("tcp", "TCP_TCC_READ_REQ_sum")
    kernels = {}
    for name in fetch:
        fb, wb = fetch[name] * 1024.0, write.get(name, 0.0) * 1024.0
        kernels[name] = dict(fetch_size_kib=fetch[name], write_size_kib=write.get(name, 0.0), launches_sampled=n[name],
                             bytes_per_launch=2.0 * fb + wb)
        if name in valu and busy.get(name):
            # cycles ONE pipe would need for every vector instruction (4 per wave64 instruction) over the cycles the kernel ran
            # (1024 SIMDs; GRBM_GUI_ACTIVE sums the 8 XCDs): above 1 means both pipes issued
            kernels[name]["valu_one_pipe_share"] = valu[name] * 4.0 / (1024.0 * busy[name] / 8.0)
        if name in hit and hit[name] + miss.get(name, 0.0) > 0:
            kernels[name]["l2_hit_rate"] = hit[name] / (hit[name] + miss[name])
        if name in l1_req and l1_req[name] > 0:
            kernels[name]["l1_hit_rate"] = 1.0 - l1_miss.get(name, 0.0) / l1_req[name]
        if name in wait and wave_cycles.get(name):
            kernels[name]["waves_waiting"] = wait[name] / wave_cycles[name]
        # the raw per-launch means bench.py derives its instruction-issue and vector-memory figures from
        raw = {}
        for pass_name, names in RAW.items():
            for cname in names:
                v = counter(pass_name, cname).get(name)
                if v is not None:
                    raw[cname] = v
        kernels[name]["per_launch"] = raw
    doc = dict(workload=bench["config"]["workload"], samples_per_dispatch=bench["config"].get("samples_per_dispatch", 1), kernels=kernels,
               note="per launch, mean over the launches of a short bench.py run under rocprofv3 --pmc (FETCH_SIZE and WRITE_SIZE in separate "
                    "passes); bytes = 2 x FETCH_SIZE (gfx950 tallies 128-byte requests as 64) + WRITE_SIZE; the doubling is an upper bound for "
                    "this kernel's scattered 64-byte reads; Infinity-Cache hits are included in FETCH_SIZE")
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
