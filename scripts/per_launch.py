#!/usr/bin/env python3
"""Per-launch durations of a rocprofv3 --kernel-trace run of scripts/render_frames.py (one frame in flight), grouped by
the launch's position within its frame: python scripts/per_launch.py <dir with *kernel_trace.csv> <launches per frame>"""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
names = ("ptr_wf_generate", "ptr_wf_extend", "ptr_wf_shade", "ptr_wf_accumulate")
rows = [r for r in rows if any(n in r[2] for n in names)]
per_frame = int(sys.argv[2])
frames = [rows[i:i + per_frame] for i in range(0, len(rows) - per_frame + 1, per_frame)][2:]   # skip two warm-up frames
for k in range(per_frame):
    d = [(fr[k][1] - fr[k][0]) / 1e3 for fr in frames]
    gap = [(fr[k][0] - fr[k - 1][1]) / 1e3 for fr in frames] if k else [0.0]
    name = [n for n in names if n in frames[0][k][2]][0]
    print(f"{k:2d} {name:20s} mean {sum(d) / len(d):8.1f} us  min {min(d):8.1f}  max {max(d):8.1f}   gap before {sum(gap) / len(gap):6.1f} us")
tot = [(fr[-1][1] - fr[0][0]) / 1e3 for fr in frames]
print(f"frame span mean {sum(tot) / len(tot):.1f} us over {len(frames)} frames")
