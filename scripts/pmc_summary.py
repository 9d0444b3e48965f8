#!/usr/bin/env python3
"""Summarises the CSVs of scripts/pmc_passes.sh: per kernel name, mean of each counter per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
rows = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"].split("(")[0]
            k = k.replace("void ptr::", "").replace("ptr::", "")
            rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(rows):
    print(k)
    for c in sorted(rows[k]):
        v = rows[k][c]
        print(f"    {c:40s} n={len(v):4d} mean={sum(v) / len(v):16.1f}")
