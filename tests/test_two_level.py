"""Two levels: a top-level tree over placed copies of one flattened tree (an extension: the reference's scene is one
triangle soup, src/scene.cpp:84-176).  The CPU restatement of the two-level traversal is checked here against the same
copies written out as one flat scene -- the GPU kernels are then checked against the restatement bit for bit."""
import numpy as np
import pytest

from common import make_bundle, oracle_frames, parse
from path_tracer_amd import host, scenes


def rotation(axis, angle):
    axis = np.asarray(axis, np.float64) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def assorted_transforms(scene, count, seed=7):
    """Rotations, non-uniform scales and one mirrored copy, spread so that the copies do not touch."""
    rng = np.random.default_rng(seed)
    P = np.asarray(scene["positions"], np.float32).reshape(-1, 3)
    size = float((P.max(axis=0) - P.min(axis=0)).max())
    m = np.zeros((count, 3, 4), np.float32)
    for k in range(count):
        A = rotation(rng.normal(size=3), rng.uniform(0, 2 * np.pi)) @ np.diag(rng.uniform(0.6, 1.5, size=3))
        if k == 1:
            A = A @ np.diag([-1.0, 1.0, 1.0])
        m[k, :, :3] = A
        m[k, :, 3] = [(k % 3 - 1) * size * 2.2, 0.3 * k, (k // 3 - 0.5) * size * 2.2]
    return m


def two_level_bundle(base, transforms, width, height, **meta_kw):
    inst = host.bvh_flatten_instances(base["flat"], transforms)
    ex = scenes.expand_instances(base, transforms)
    lo, hi = ex["positions"].reshape(-1, 3).min(axis=0), ex["positions"].reshape(-1, 3).max(axis=0)
    centre, size = (lo + hi) / 2, float((hi - lo).max())
    pos = centre + np.array([0.3, 0.5, -0.6]) * size
    cam = host.camera(pos, centre, 60.0, width / height, 0.0, float(np.linalg.norm(pos - centre)))
    meta = scenes.pack_metadata(cam, width, height, **meta_kw)
    two = dict(base, flat_instances=inst, meta=meta, width=width, height=height)
    nodes = host.bvh_build(ex["triangles"], mode=1)
    flat = host.bvh_flatten(ex["triangles"], None)
    expanded = dict(ex, nodes=nodes, flat=flat, meta=meta, width=width, height=height)
    return two, expanded


def instance_boxes(blob):
    """(instance index -> decoded leaf box) and the number of times each instance is referenced."""
    h, nodes, _ = parse(blob)
    ni, nu = nodes.view(np.int32), nodes.view(np.uint32)
    boxes = {}

    def walk(ref, lo, hi):
        if ref < 0:
            r = ~ref
            assert r & 0x40000000
            index = r & 0x3FFFFFFF
            assert index not in boxes
            boxes[index] = (lo, hi)
            return
        nd = nodes[ref]
        step = np.array([nd[3], nd[14], nd[15]], np.float32)
        for k in range(4):
            child = int(ni[ref, 4 + k])
            if child == -2 ** 31:
                continue
            qlo = np.array([(int(nu[ref, 8 + a]) >> (8 * k)) & 0xFF for a in range(3)], np.float32)
            qhi = np.array([(int(nu[ref, 11 + a]) >> (8 * k)) & 0xFF for a in range(3)], np.float32)
            walk(child, nd[0:3] + qlo * step, nd[0:3] + qhi * step)

    walk(h["root"], h["scene_min"], h["scene_max"])
    return h, boxes


def test_instance_layout():
    base = make_bundle("tiny", 0, 32, 24)
    m = assorted_transforms(base, 7)
    blob = host.bvh_flatten_instances(base["flat"], m)
    h, boxes = instance_boxes(blob)
    assert h["magic"] == 0x49525450 and h["triangle_count"] == 7 and sorted(boxes) == list(range(7))
    oh = host.flat_header(base["flat"])
    assert h["max_depth"] >= oh["max_depth"] + 1
    rec = blob[h["triangles_offset"]:h["triangles_offset"] + 64 * 7].view(np.float32).reshape(7, 16)
    P = np.asarray(base["positions"], np.float64).reshape(-1, 3)
    for k in range(7):
        W = rec[k, :12].reshape(3, 4).astype(np.float64)
        A, t = m[k, :, :3].astype(np.float64), m[k, :, 3].astype(np.float64)
        assert np.allclose(W[:, :3] @ A, np.eye(3), atol=1e-5) and np.allclose(W[:, :3] @ t + W[:, 3], 0, atol=1e-4)
        assert rec.view(np.int32)[k, 12] == oh["root"]
        world = P @ A.T + t
        lo, hi = boxes[k]
        assert (lo <= world.min(axis=0)).all() and (hi >= world.max(axis=0)).all(), "the instance's box holds its geometry"


def test_instance_build_rejects_bad_input():
    base = make_bundle("tiny", 0, 32, 24)
    good = scenes.grid_transforms(base, 2, 2)
    bad = good.copy(); bad[1, :, :3] = 0.0
    with pytest.raises(host.HostError, match="singular"):
        host.bvh_flatten_instances(base["flat"], bad)
    bad = good.copy(); bad[2, 0, 3] = np.inf
    with pytest.raises(host.HostError, match="non-finite"):
        host.bvh_flatten_instances(base["flat"], bad)
    with pytest.raises(host.HostError, match="not a flattened tree"):
        host.bvh_flatten_instances(host.bvh_flatten_instances(base["flat"], good), good)


@pytest.mark.parametrize("which", ["grid", "assorted"])
def test_two_level_matches_the_copies_written_out(which):
    base = make_bundle("tiny", 0, 32, 24)
    m = scenes.grid_transforms(base, 3, 2) if which == "grid" else assorted_transforms(base, 6)
    two, expanded = two_level_bundle(base, m, 96, 64, max_bounce=3)
    img2, c2, cap2 = oracle_frames(two, 2, use_flat=True, aov=True)
    img1, c1, cap1 = oracle_frames(expanded, 2, use_flat=True, aov=True)
    hit_fraction = (cap1[0][..., 0] >= 0).mean()
    assert hit_fraction > 0.1, "the camera sees the copies"
    same = cap1[0][..., 0] == cap2[0][..., 0]
    assert same.mean() > 0.995, "primary hits name the same triangle of the same copy"
    t1, t2 = cap1[1][..., 0][same], cap2[1][..., 0][same]
    assert np.allclose(t1, t2, rtol=2e-4, atol=1e-5)
    assert np.abs(img1[..., :3] - img2[..., :3]).mean() < (0.004 if which == "grid" else 0.02)
    assert c2["nodes"] > 0 and c2["triangles"] > 0


def test_single_instance_identity_is_the_object():
    """One copy under the identity matrix: the ray passes through unchanged, so primary hits keep their bits; the normals
    are normalised once more on the way back to world space, so later bounces agree to rounding only."""
    base = make_bundle("tiny", 0, 48, 32, max_bounce=3)
    ident = np.zeros((1, 3, 4), np.float32); ident[0, 0, 0] = ident[0, 1, 1] = ident[0, 2, 2] = 1.0
    two = dict(base, flat_instances=host.bvh_flatten_instances(base["flat"], ident))
    img2, _, cap2 = oracle_frames(two, 2, use_flat=True, aov=True)
    img1, _, cap1 = oracle_frames(base, 2, use_flat=True, aov=True)
    assert np.array_equal(cap1[0][..., 0], cap2[0][..., 0]) and np.array_equal(cap1[1][..., 0], cap2[1][..., 0])
    assert (cap1[0][..., 1] == cap2[0][..., 1]).mean() > 0.99
    assert np.abs(img1 - img2).mean() < 2e-3


def two_object_bundle(width, height, copies=7, **meta_kw):
    """Two different objects (the tiny scene and a small atrium) in one layout, placed `copies` times in turn."""
    a = scenes.tiny_scene(3)
    b = scenes.generate("atrium", 1500, seed=5, texture_size=32)
    # bring the atrium to the tiny scene's size so that both show up in one view
    pa = np.asarray(a["positions"], np.float32).reshape(-1, 3)
    pb = np.asarray(b["positions"], np.float32).reshape(-1, 3)
    scale = np.float32((pa.max(axis=0) - pa.min(axis=0)).max() / (pb.max(axis=0) - pb.min(axis=0)).max())
    b = dict(b, positions=(pb * scale).reshape(-1), triangles=np.asarray(b["triangles"], np.float32) * scale)
    sc, object_first = scenes.concat_scenes([a, b])
    flat, objects = host.bvh_flatten_objects(sc["triangles"], object_first)
    which = np.arange(copies, dtype=np.uint32) % 2
    m = assorted_transforms(a, copies, seed=11)
    top = host.bvh_flatten_placements(objects, which, m)
    # camera over the placements
    centre = m[:, :, 3].mean(axis=0).astype(np.float64)
    size = float(np.ptp(m[:, :, 3], axis=0).max()) + 2.0
    pos = centre + np.array([0.3, 0.5, -0.6]) * size
    cam = host.camera(pos, centre, 60.0, width / height, 0.0, float(np.linalg.norm(pos - centre)))
    nodes = np.zeros((1, 8), np.uint32); nodes[0, 3] = 0xFFFFFFFF      # the threaded tree is not walked
    return dict(sc, nodes=nodes, flat=flat, flat_instances=top, meta=scenes.pack_metadata(cam, width, height, **meta_kw), width=width, height=height,
                object_first=object_first, object_of_copy=which, objects=objects)


def test_objects_share_one_layout():
    b = two_object_bundle(32, 24)
    h, nodes, tris = parse(b["flat"])
    first = b["object_first"]
    assert h["triangle_count"] == first[-1] and sorted(tris.view(np.int32)[:, 7].tolist()) == list(range(first[-1])), "primitive ids are global"
    assert h["root"] == b["objects"]["root"][0] and h["max_depth"] == b["objects"]["max_depth"].max()
    # each object's tree reaches exactly its own triangles
    ni = nodes.view(np.int32)

    def reach(ref, out):
        if ref < 0:
            r = ~ref
            out.update(range(r >> 2, (r >> 2) + (r & 3) + 1))
            return
        for k in range(4):
            c = int(ni[ref, 4 + k])
            if c != -2 ** 31:
                reach(c, out)

    seen_all = set()
    for k, o in enumerate(b["objects"]):
        got = set()
        reach(int(o["root"]), got)
        prims = set(tris.view(np.int32)[sorted(got), 7].tolist())
        assert prims == set(range(int(first[k]), int(first[k + 1]))), f"object {k}"
        assert not (got & seen_all)
        seen_all |= got
        tri = b["triangles"][first[k]:first[k + 1]].reshape(-1, 3)
        assert np.array_equal(o["box_min"], tri.min(axis=0)) and np.array_equal(o["box_max"], tri.max(axis=0))


def test_copies_of_two_objects():
    """Every copy shows the object it names: a primary hit's primitive lies in that object's run of triangles."""
    b = two_object_bundle(96, 64, max_bounce=3)
    img, counters, cap = oracle_frames(b, 2, use_flat=True, aov=True)
    named = cap[0][..., 0]
    hit = named >= 0
    assert hit.mean() > 0.05
    total = int(b["object_first"][-1])
    copy, prim = named[hit] // total, named[hit] % total
    assert len(np.unique(copy)) >= 4
    obj = b["object_of_copy"][copy]
    assert ((prim >= b["object_first"][obj]) & (prim < b["object_first"][obj + 1])).all()
    assert set(np.unique(obj).tolist()) == {0, 1}, "both objects are in view"
    assert np.isfinite(img).all() and img[..., :3].std() > 0.01
