#!/usr/bin/env python3
"""Nodes and triangles per ray of the flattened layout on the bench workload, counted by the CPU oracle on bands of rows
spread over the frame: what a change to the builder is worth before any GPU time is spent on it (the traversal kernel is
bound by instruction issue, so its time follows 125 x nodes + 66 x triangles per ray).  Kept beside the tests because it runs the
oracle; pytest does not collect it."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from path_tracer_amd import host, scenes
import pyoracle

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="street_exterior")
ap.add_argument("--triangles", type=int, default=2_800_000)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--bounces", type=int, default=4)
ap.add_argument("--bands", type=int, default=24)
ap.add_argument("--rows", type=int, default=4)
ap.add_argument("--samples", type=int, default=2)
a = ap.parse_args()
pyoracle.build()
t0 = time.time()
sc = scenes.generate(a.scene, a.triangles, texture_size=64)
t1 = time.time()
flat = host.bvh_flatten(sc["triangles"], None, threads=os.cpu_count())
t2 = time.time()
c = sc["camera"]
cam = host.camera(c["pos"], c["target"], c["v_fov"], a.width / a.height, c["aperture"], c["focus"])
meta = scenes.pack_metadata(cam, a.width, a.height, max_bounce=a.bounces, min_bounce=1)
nodes = np.zeros((1, 8), np.uint32); nodes[0, 3] = 0xFFFFFFFF
o = pyoracle.OracleScene(dict(sc, meta=meta, nodes=nodes, flat=flat))
acc = np.zeros((a.height, a.width, 4), np.float32); out = np.zeros_like(acc)
tot = dict(rays=0, nodes=0, triangles=0)
step = (a.height - a.rows) // max(1, a.bands - 1)
for s in range(1, a.samples + 1):
    o.set_meta(sample_index=s)
    for b in range(a.bands):
        y = b * step
        r = o.render(acc, out, rows=(y, y + a.rows), threads=os.cpu_count(), use_flat=True)
        for k in tot: tot[k] += r[k]
n, t = tot["nodes"] / tot["rays"], tot["triangles"] / tot["rays"]
print(f"triangles {sc['triangles'].shape[0]}  flat bytes {len(flat) if hasattr(flat, '__len__') else 0}  build {t2 - t1:.1f}s  rays {tot['rays']}  "
      f"nodes/ray {n:.3f}  tris/ray {t:.3f}  valu/ray {125 * n + 66 * t:.0f}")
