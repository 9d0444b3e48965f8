"""C1 of the measurement configurations: the Sponza-class atrium at 1280x720, one sample, max_bounce 4, on the CPU alone --
the reference-order BVH build and the oracle on all host cores (the configuration the reference itself could run without a
GPU).  A test, not a benchmark: it checks that the configuration runs at its real size and prints the two times."""
import os
import time

import numpy as np

from common import make_bundle, oracle_frames


def test_c1_atrium_720p_on_the_cpu(capsys):
    t0 = time.time()
    b = make_bundle("atrium", 262144, 1280, 720, max_bounce=4)
    t_build = time.time() - t0                                    # generator + reference-order tree + ranks + flattened layout
    n = b["triangles"].shape[0]
    assert 250_000 < n < 275_000 and b["nodes"].shape[0] == 2 * n - 1, "the reference's builder emits 2N - 1 threaded nodes"
    t0 = time.time()
    img, counters, _ = oracle_frames(b, 1, threads=os.cpu_count() or 1)
    t_render = time.time() - t0
    assert counters["nan_hits"] == 0
    assert counters["rays"] >= 1280 * 720 and counters["shaded"] > 0.3 * 1280 * 720, "most primary rays hit the atrium"
    assert np.isfinite(img).all() and (img[..., 3] == 1.0).all() and 0.0 <= img[..., :3].min() and img[..., :3].max() <= 1.0
    assert img[..., :3].std() > 0.05, "a picture, not a flat colour"
    with capsys.disabled():
        print(f"\n[C1] {n} triangles: scene + trees {t_build:.1f} s; 1280x720, 1 spp, 4 bounces on {os.cpu_count()} threads: "
              f"{t_render:.1f} s, {counters['rays'] / t_render / 1e6:.2f} Mrays/s, {counters['nodes'] / counters['rays']:.1f} nodes per ray "
              f"(the reference's threaded tree)")
