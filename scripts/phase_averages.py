#!/usr/bin/env python3
"""Average launch duration of the traversal kernel per phase of one `bench.py` run, from rocprofv3's kernel trace of that
run: the trace's own --stats average mixes the warm-up launches (the very first one pays the first touch of every buffer), the
timed region's (two dispatches in flight) and the solo leg's (one in flight); the bench line reports the last two.
   usage: scripts/phase_averages.py <..._kernel_trace.csv> <bench line .json>"""
import csv, json, sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "ptr_wf_extend<false" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ms = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
line = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
timed = line["roofline"]["launches"]
# launch order of the run: warm-up, the timed region, the solo leg, then whatever else the run does (the one-sample legs unless
# PTR_BENCH_SKIP_SINGLE is set, the parity check's one-sample frames unless --skip-parity is given)
cfg = line["config"]
per_dispatch = timed // max(1, cfg["dispatches_per_gpu"])
warm = -(-cfg["warmup_steps_run"] // cfg["samples_per_dispatch"]) * per_dispatch
solo = timed
mean = lambda v: sum(v) / max(1, len(v))
print(f"kernel ptr_wf_extend<false, ...>: {len(ms)} launches in the trace, all: {mean(ms):.4f} ms")
print("per launch, ms:", " ".join(f"{x:.3f}" for x in ms))
print(f"warm-up   ({warm:3d} launches): {mean(ms[:warm]):.4f} ms")
print(f"timed     ({timed:3d} launches): {mean(ms[warm:warm + timed]):.4f} ms   bench line roofline.avg_launch_ms      = {line['roofline']['avg_launch_ms']:.4f}")
print(f"solo      ({solo:3d} launches): {mean(ms[warm + timed:warm + timed + solo]):.4f} ms   bench line roofline.solo_avg_launch_ms = {line['roofline']['solo_avg_launch_ms']:.4f}")
rest = ms[warm + timed + solo:]
print(f"afterwards ({len(rest):3d} launches, one sample each): {mean(rest):.4f} ms")
