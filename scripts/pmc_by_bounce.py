#!/usr/bin/env python3
"""Per-bounce view of the extend kernel's counters from a scripts/pmc_passes.sh output directory."""
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
per = defaultdict(lambda: defaultdict(list))   # bounce -> counter -> values
dur = defaultdict(list)
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "wf_extend" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    order = {d: i % 4 for i, d in enumerate(ids)}
    for r in rows:
        per[order[int(r["Dispatch_Id"])]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(os.path.join(out, "sq1", "**", "*kernel_trace.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "wf_extend" in r["Kernel_Name"]]
    for i, r in enumerate(sorted(rows, key=lambda r: int(r["Dispatch_Id"]))):
        dur[i % 4].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
names = sorted({c for b in per.values() for c in b})
print(f"{'counter':36s}" + "".join(f"{'bounce ' + str(b):>16s}" for b in range(4)))
print(f"{'duration_us (sq1 pass)':36s}" + "".join(f"{sum(dur[b]) / max(1, len(dur[b])):16.1f}" for b in range(4)))
for c in names:
    print(f"{c:36s}" + "".join(f"{sum(per[b][c]) / max(1, len(per[b][c])):16.0f}" for b in range(4)))
