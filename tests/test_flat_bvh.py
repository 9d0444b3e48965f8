"""The flattened GPU layout emitted by bvh::flatten(), checked on the CPU: structural invariants, and its
traversal (as restated in the oracle) against the reference-order traversal on the same rays."""
import ctypes as C

import numpy as np

import pyoracle
from common import make_bundle, oracle_frames
from path_tracer_amd import host


from common import check_flat_layout, parse  # noqa: E402,F401


def test_layout_invariants():
    b = make_bundle("atrium", 6000)
    h = check_flat_layout(b["flat"], b["triangles"], b["ranks"])
    assert b["flat"].nbytes == 64 + 64 * (h["node_count"] + b["triangles"].shape[0])


def test_tiny_inputs_flatten():
    rng = np.random.default_rng(5)
    for n in (1, 2, 3, 4, 5, 9):
        tri = rng.random((n, 9), dtype=np.float32)
        h, nodes, tris = parse(host.bvh_flatten(tri))
        assert h["triangle_count"] == n and (h["root"] < 0) == (h["node_count"] == 0)
        assert sorted(tris.view(np.int32)[:, 7].tolist()) == list(range(n))


def test_flat_traversal_agrees_with_reference_order_traversal_on_random_rays():
    b = make_bundle("atrium", 6000)
    o = pyoracle.OracleScene(b)
    acc = np.zeros((b["height"], b["width"], 4), np.float32)
    f = o.frame(acc, acc.copy())
    L = pyoracle.lib()
    rng = np.random.default_rng(11)
    lo, hi = host.flat_header(b["flat"])["scene_min"], host.flat_header(b["flat"])["scene_max"]
    mismatches = 0
    c_ref, c_flat = pyoracle.Counters(), pyoracle.Counters()
    for _ in range(3000):
        org = (lo + (hi - lo) * rng.random(3)).astype(np.float32)
        d = rng.normal(size=3).astype(np.float32)
        pr, pf = C.c_int32(-1), C.c_int32(-1)
        tr, tf = C.c_float(), C.c_float()
        br, bf = np.zeros(3, np.float32), np.zeros(3, np.float32)
        hr = L.orc_closest_hit(C.byref(f), org.ctypes.data, d.ctypes.data, 0.001, 1e15, C.byref(pr), C.byref(tr), br.ctypes.data, C.byref(c_ref))
        hf = L.orc_closest_hit_flat(b["flat"].ctypes.data, org.ctypes.data, d.ctypes.data, 0.001, 1e15, C.byref(pf), C.byref(tf), bf.ctypes.data, C.byref(c_flat))
        if hr != hf or (hr and (pr.value != pf.value or tr.value != tf.value or not np.array_equal(br, bf))):
            mismatches += 1
    assert mismatches == 0
    assert c_flat.nodes * 8 < c_ref.nodes   # the point of the exercise: far fewer node fetches per ray


def test_flat_and_reference_order_images_agree():
    b = make_bundle("atrium", 5000, width=96, height=54, max_bounce=4)
    ref, c_ref, _ = oracle_frames(b, 2)
    flat, c_flat, _ = oracle_frames(b, 2, use_flat=True)
    differing = int((np.abs(ref - flat).max(axis=2) > 0).sum())
    assert differing <= 1   # only reference box-rounding cracks may differ; none expected at this size
    assert c_ref["rays"] == c_flat["rays"] or differing > 0
