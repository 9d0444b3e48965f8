"""The flattened GPU layout emitted by bvh::flatten(), checked on the CPU: structural invariants, and its
traversal (as restated in the oracle) against the reference-order traversal on the same rays."""
import ctypes as C

import numpy as np

import pyoracle
from common import make_bundle, oracle_frames
from path_tracer_amd import host


def parse(flat):
    h = host.flat_header(flat)
    nodes = flat[h["nodes_offset"]:h["nodes_offset"] + 64 * h["node_count"]].view(np.float32).reshape(-1, 16)
    tris = flat[h["triangles_offset"]:h["triangles_offset"] + 64 * h["triangle_count"]].view(np.float32).reshape(-1, 16)
    return h, nodes, tris


def test_layout_invariants():
    b = make_bundle("atrium", 6000)
    h, nodes, tris = parse(b["flat"])
    n = b["triangles"].shape[0]
    assert h["magic"] == 0x46525450 and h["version"] == 2 and h["triangle_count"] == n
    assert h["nodes_offset"] % 64 == 0 and h["triangles_offset"] % 64 == 0 and b["flat"].nbytes == 64 + 64 * (h["node_count"] + n)
    ti = tris.view(np.int32)
    assert sorted(ti[:, 7].tolist()) == list(range(n))                        # every primitive exactly once
    assert np.array_equal(ti[:, 3], b["ranks"][ti[:, 7]])                      # rank travels with the primitive
    tri = b["triangles"].reshape(n, 3, 3)[ti[:, 7]]
    assert np.array_equal(tris[:, 0:3], tri[:, 0])                             # p0, e1, e2, n = cross(e1, e2) in fp32
    e1, e2 = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
    assert np.array_equal(tris[:, 4:7], e1) and np.array_equal(tris[:, 8:11], e2)
    nrm = np.stack([e1[:, 1] * e2[:, 2] - e2[:, 1] * e1[:, 2], e1[:, 2] * e2[:, 0] - e2[:, 2] * e1[:, 0],
                    e1[:, 0] * e2[:, 1] - e2[:, 0] * e1[:, 1]], 1)
    assert np.array_equal(tris[:, 12:15], nrm)

    # walk the 4-wide tree: every decoded child box contains its subtree (conservative quantisation, within one
    # step of the true bound), leaves hold 1..4 triangles and tile the triangle array, the stack bound holds
    ni = nodes.view(np.int32)
    nu = nodes.view(np.uint32)
    seen = np.zeros(n, bool)
    tmin, tmax = tri.min(axis=1), tri.max(axis=1)
    EMPTY = -2 ** 31
    worst_stack = 0

    def bounds_of(ref, pending):
        nonlocal worst_stack
        if ref < 0:
            first, cnt = (~ref) >> 2, ((~ref) & 3) + 1
            assert not seen[first:first + cnt].any()
            seen[first:first + cnt] = True
            return tmin[first:first + cnt].min(axis=0), tmax[first:first + cnt].max(axis=0)
        nd = nodes[ref]
        count = sum(1 for k in range(4) if int(ni[ref, 4 + k]) != EMPTY)
        assert 2 <= count <= 4
        step = np.array([nd[3], nd[14], nd[15]], dtype=np.float32)
        assert all(float(x) > 0 and np.frexp(x)[0] == 0.5 for x in step), "steps are powers of two"
        worst_stack = max(worst_stack, pending + count - 1)
        lo_all, hi_all = [], []
        for k in range(4):
            child = int(ni[ref, 4 + k])
            if k >= count:
                assert child == EMPTY
                continue
            assert child != EMPTY
            qlo = np.array([(int(nu[ref, 8 + a]) >> (8 * k)) & 0xFF for a in range(3)], dtype=np.float32)
            qhi = np.array([(int(nu[ref, 11 + a]) >> (8 * k)) & 0xFF for a in range(3)], dtype=np.float32)  # words 11, 12, 13
            dlo = nd[0:3] + qlo * step
            dhi = nd[0:3] + qhi * step
            lo, hi = bounds_of(child, pending + count - 1)
            assert (dlo <= lo).all() and (dhi >= hi).all(), "decoded box must contain the subtree"
            assert (lo - dlo <= step * 2.001 + 1e-30).all() and (dhi - hi <= step * 2.001 + 1e-30).all(), "and be tight to two steps"
            lo_all.append(lo); hi_all.append(hi)
        lo, hi = np.min(lo_all, axis=0), np.max(hi_all, axis=0)
        slack = (hi - lo) * 2e-6 + 2 * np.spacing(np.abs(lo).astype(np.float32)) + 1e-29
        assert (nd[0:3] < lo).all() and (lo - nd[0:3] <= slack).all(), "node origin sits a margin below the true box minimum"
        return lo, hi

    import sys
    sys.setrecursionlimit(10000)
    lo, hi = bounds_of(h["root"], 0)
    assert seen.all() and worst_stack <= h["max_depth"] <= 62
    assert np.array_equal(lo, h["scene_min"]) and np.array_equal(hi, h["scene_max"])


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
