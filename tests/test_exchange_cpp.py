"""The C++ side's multi-GPU exchange on the CPU: several processes, each one rank of host/exchange_selftest -- the stripe
layout of the tile partition (pack, all-gather, unpack) and the all-reduce sum over the file transport.  (The same layout on
device memory and the RCCL transport run in tests/test_gpu_headless_cpp.py.)"""
import os
import subprocess

import pytest

from path_tracer_amd import build, paths

SELFTEST = os.path.join(paths.HOST, "exchange_selftest")


@pytest.mark.parametrize("world,width,height,stripe", [(2, 40, 37, 8), (3, 17, 64, 8), (4, 33, 50, 16), (1, 8, 8, 8)])
def test_tile_gather_and_sum_over_processes(tmp_path, world, width, height, stripe):
    if not os.path.exists(SELFTEST):
        build.build_headless()
    procs = [subprocess.Popen([SELFTEST, str(tmp_path), str(r), str(world), str(width), str(height), str(stripe)], stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    for r, p in enumerate(procs):
        _, err = p.communicate(timeout=120)
        assert p.returncode == 0, f"rank {r}: {err}"


def test_a_missing_rank_is_reported_not_waited_for_forever(tmp_path):
    """Rank 1 of 2 never starts: rank 0 gives up with a message naming the file it waited for (the time limit is the
    transport's own, two minutes; here the process is stopped after a few seconds and must still be waiting, not crashed)."""
    p = subprocess.Popen([SELFTEST, str(tmp_path), "0", "2", "16", "16", "8"], stderr=subprocess.PIPE, text=True)
    try:
        p.communicate(timeout=3)
        raise AssertionError("rank 0 finished without its peer")
    except subprocess.TimeoutExpired:
        p.kill()
        p.communicate()
    assert os.path.exists(os.path.join(str(tmp_path), "gather_0_0.bin"))     # its own contribution was published, complete
