"""The multi-GPU exchange steps on the CPU: two gloo ranks.  The render on each rank is done by the CPU oracle
(no GPU here); what is under test is the partition arithmetic and the collectives of path_tracer_amd.distributed."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from common import make_bundle, oracle_frames


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, stripe, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.join(here, "..", "path-tracer_amd", "python"), os.path.join(here, "..", "oracle"), here):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pyoracle
        from path_tracer_amd import distributed as ptdist
        b = make_bundle("tiny", 0, 40, 37, max_bounce=3)          # 37 rows: not a multiple of the stripe
        H, W = b["height"], b["width"]
        full, _, _ = oracle_frames(b, 2, threads=1)

        # tile partition: render only the owned stripes (two samples, ping-pong), then one all-gather
        o = pyoracle.OracleScene(b)
        acc = np.zeros((H, W, 4), np.float32); out = np.zeros_like(acc)
        stripes = ptdist.owned_stripes(rank, world, stripe, H)
        for s in (1, 2):
            o.set_meta(sample_index=s)
            for st in stripes:
                o.render(acc, out, rows=(st * stripe, min(H, (st + 1) * stripe)), threads=1)
            acc, out = out, acc
        mask = ptdist.owned_row_mask(rank, world, stripe, H).numpy()
        assert (acc[~mask] == 0).all()
        merged = ptdist.gather_tiles(torch.from_numpy(acc), rank, world, stripe).numpy()
        ok_tiles = bool(np.array_equal(merged, full))

        # sample partition: sum of per-rank linear images with alpha = contribution count
        lin = torch.full((H, W, 4), float(rank + 1))
        lin[..., 3] = 1.0
        ptdist.allreduce_linear(lin)
        ok_sum = bool((lin[..., :3] == sum(range(1, world + 1))).all() and (lin[..., 3] == world).all())
        q.put((rank, ok_tiles, ok_sum))
    finally:
        dist.destroy_process_group()


def test_tile_gather_and_sample_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 8, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True, True), (1, True, True)]


def test_partition_arithmetic():
    from path_tracer_amd import distributed as ptdist
    for world in (1, 2, 3, 8):
        for H in (8, 37, 1080):
            owned = [ptdist.owned_stripes(r, world, 8, H) for r in range(world)]
            allst = sorted(s for o in owned for s in o)
            assert allst == list(range((H + 7) // 8))
            masks = torch.stack([ptdist.owned_row_mask(r, world, 8, H) for r in range(world)])
            assert (masks.sum(dim=0) == 1).all()
    # single-rank gather is the identity
    img = torch.rand(19, 5, 4)
    assert torch.equal(ptdist.gather_tiles(img, 0, 1, 8), img)
