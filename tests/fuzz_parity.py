#!/usr/bin/env python3
"""Randomised parity sweep: random scenes, frame sizes, metadata and dispatch options through the C ABI against the CPU oracle,
image for image (numpy ==).  Kept beside the tests (it checks against the oracle, as they do) but not collected by pytest.  What the fixed test cases do not reach by construction -- odd sizes together with partitions,
held-back dispatches together with several samples per dispatch, deep bounce counts with the cooperative last-ray walk, ...
   python tests/fuzz_parity.py --iterations 200 --seed 1      (on a GPU box; prints one line per case, stops at the first mismatch)"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import pyoracle
    from common import add_spheres, make_bundle, oracle_frames
    from path_tracer_amd import scenes
    from test_two_level import assorted_transforms, two_level_bundle
    from path_tracer_amd import hip
    pyoracle.build()
    rng = np.random.default_rng(args.seed)
    ctx = hip.Context(0)
    t0 = time.time()
    for it in range(args.iterations):
        if it and it % 20 == 0:   # a context never gives image-table entries back: start afresh now and then
            ctx.close()
            ctx = hip.Context(0)
        kind = str(rng.choice(["tiny", "atrium", "street_interior", "street_exterior"]))
        target = int(rng.integers(500, 40_000))
        W, H = int(rng.integers(9, 200)), int(rng.integers(9, 140))
        meta_kw = dict(max_bounce=int(rng.integers(1, 7)), min_bounce=int(rng.integers(0, 3)))
        if rng.random() < 0.3:
            meta_kw["enable_dof"] = 1
        if rng.random() < 0.25:
            meta_kw["downscale_factor"] = int(rng.integers(2, 4))
        b = make_bundle(kind, target, W, H, seed=int(rng.integers(1, 1 << 30)), **meta_kw)
        tree = str(rng.choice(["host", "host", "gpu", "copies", "spheres"]))
        if tree == "spheres":     # sphere leaves among the triangles: anywhere in the scene's box, any material the scene could have
            lo, hi = b["triangles"].reshape(-1, 3).min(axis=0), b["triangles"].reshape(-1, 3).max(axis=0)
            extent = float((hi - lo).max())
            eye = np.float32(b["camera"]["pos"])
            spheres = []
            for _ in range(int(rng.integers(1, 60))):
                m = np.zeros((), dtype=scenes.MATERIAL_DTYPE)
                m["base_color"] = list(rng.random(3)) + [1.0]
                m["metalness"], m["roughness"] = [(0.0, 1.0), (1.0, 0.0), (1.0, float(rng.uniform(0.1, 0.9))), (float(rng.random()), float(rng.random()))][int(rng.integers(0, 4))]
                if rng.random() < 0.3 and len(b["textures"]) > 1:
                    m["albedo_tex"] = int(rng.integers(1, len(b["textures"])))
                near = rng.random() < 0.4   # some of them around the camera, so that they are seen at any scene size
                centre = (eye + rng.normal(size=3) * 0.08 * extent) if near else rng.uniform(lo, hi)
                spheres.append((tuple(np.float32(centre)), float(rng.uniform(0.002, 0.06) * extent), m))
            b = add_spheres(b, spheres)
        if tree == "copies":      # the scene placed several times through a top-level tree (rotated, scaled, one mirrored)
            b, _ = two_level_bundle(b, assorted_transforms(b, int(rng.integers(1, 7)), seed=int(rng.integers(1, 1000))), W, H, **meta_kw)
        spp = int(rng.integers(1, 8))
        per = int(rng.integers(1, 5))
        hold = int(rng.choice([1, 1, 2, 3, 8]))
        # now and then the strict drop-in kernel over the reference's own tree (one sample per dispatch, nothing held back),
        # against the oracle's walk of that tree
        mega = tree == "host" and rng.random() < 0.2
        if mega:
            per, hold = 1, 1
        in_flight = int(rng.choice([1, 2, 2, 3]))
        world = int(rng.choice([1, 1, 2, 3]))
        rank = int(rng.integers(0, world))
        stripe = int(rng.choice([8, 16, 24]))
        refill = int(rng.choice([8, 16, 32, 64]))
        alone = int(rng.choice([0, 0, 3, 1000]))
        what = (f"{kind}/{'mega' if mega else tree} {b['triangles'].shape[0]} tris {W}x{H} {meta_kw} spp {spp} per-dispatch {per} hold {hold} in-flight {in_flight} "
                f"rank {rank}/{world} stripe {stripe} refill {refill} alone {alone}")
        gs = hip.GpuScene(ctx, b, b["nodes"], b["flat"], b["meta"])
        if tree == "gpu":         # the layout built on the GPU from the uploaded buffers; the oracle walks that very blob
            built = ctx.build_flat(gs.indices_buffer, gs.positions_buffer, b["triangles"].shape[0])
            ctx.bind_flat(gs.bvh_buffer, built)
            b = dict(b, flat=built.read(np.uint8))
        ctx.set_option("frames.in_flight", in_flight)
        ctx.set_option("extend.refill", refill)
        ctx.set_option("extend.alone_turns", alone)
        ctx.set_option("dispatch.coalesce", hold)
        ctx.set_partition(rank, world, stripe)
        # downscaled frames write blocks: the image is as large as the blocks reach
        loop = hip.FrameLoop(ctx, gs, W, H, "compute.megakernel" if mega else "compute", samples_per_dispatch=per)
        try:
            loop.frames(spp)
            img = loop.latest().read()
        finally:
            ctx.set_option("dispatch.coalesce", 1)
            ctx.set_partition(0, 1, 8)
        ref, counters, _ = oracle_frames(b, spp, use_flat=not mega, materials=gs.materials_host, textures=gs.texture_table,
                                         spheres=gs.spheres_host)
        d = meta_kw.get("downscale_factor", 1)
        # a rank owns rows of the DISPATCH domain (frame rows / downscale); its image rows follow from the block size
        owned = np.zeros(H, bool)
        for gy in range((H + d - 1) // d):
            if world == 1 or (gy // stripe) % world == rank:
                owned[gy * d:(gy + 1) * d] = True
        ok = np.array_equal(img[owned], ref[owned]) and (world == 1 or (img[~owned] == 0).all())
        print(f"[{it:4d}] {'ok ' if ok else 'MISMATCH'} {what}", flush=True)
        if not ok:
            bad = (img[owned] != ref[owned]).any(axis=-1).sum()
            print(f"       {int(bad)} owned pixels differ; foreign rows written: {int((img[~owned] != 0).any(axis=-1).sum())}")
            sys.exit(1)
        for t in (loop.accumulation, loop.output):
            t.destroy()
    print(f"{args.iterations} cases equal in {time.time() - t0:.0f}s")
    ctx.close()


if __name__ == "__main__":
    main()
