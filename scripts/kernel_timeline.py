#!/usr/bin/env python3
"""The last N kernel launches of a rocprofv3 --kernel-trace run as a timeline (start, end, duration in us, queue, name):
shows which launches of the two dispatches in flight ran beside each other and what ran alone at the edges of a region.
   usage: scripts/kernel_timeline.py <rocprofv3 output dir> <N>"""
import csv, glob, sys

trace = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))[-int(sys.argv[2]):]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    start, end = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{start:9.1f} {end:9.1f} {end - start:8.1f} us  queue {r.get('Queue_Id', '?')}  {r['Kernel_Name'][:64]}")
