"""The scene-loading step in front of the render path (glTF -> flat arrays; src/gltf.cpp, src/mesh.cpp,
src/scene.cpp:84-176 in the reference): JSON reader, PNG / JPEG decoders and the flattening, all CPU code of
libptr_host.so.  The reference's loader cannot be built here (it includes the Vulkan renderer headers) and its model
files are Git-LFS pointers, so the expected values are the arrays the files were written from (round trip through
independent Python writers), Python's json / zlib / Pillow as second opinions, and hand-made files for the format's
corner cases."""
import base64
import io
import json
import os
import struct

import numpy as np
import pytest

from path_tracer_amd import gltf_io, host, scenes


# ---- JSON ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("text", [
    '{"a":[1,2.5,-3e2,0.1,1e-7,123456789012],"b":{"c":null,"d":true,"e":false},"s":"x\\ny\\u00e9\\ud83d\\ude00\\"\\\\\\/"}',
    '  [ ]  ', '{}', '[[[[1]]]]', '"just a string"', '-0.0', '1E+2', '{"k":1,"k":2}',
    '﻿{"bom": 1}',
])
def test_json_reader_agrees_with_python(text):
    got = json.loads(host.json_dump(text))
    want = json.loads(text.lstrip("﻿"))
    assert got == want      # an integral float prints without its fraction: equal as numbers


def test_json_numbers_survive_bit_exactly():
    rng = np.random.default_rng(0)
    vals = [float(np.float32(v)) for v in rng.standard_normal(200) * 10.0 ** rng.integers(-20, 20, 200)]
    got = json.loads(host.json_dump(json.dumps(vals)))
    assert np.array_equal(np.array(got, np.float64), np.array(vals, np.float64))
    assert np.array_equal(np.array(got, np.float64).astype(np.float32), np.array(vals, np.float32))


@pytest.mark.parametrize("text", ['{"a":1,}', '[1 2]', '{"a" 1}', '01', '1.', '.5', '+1', 'nul', '"unterminated', '"bad \\x escape"',
                                  '"\\ud800 lone"', '[1] trailing', '', '{"a":\t"raw\ttab ok"}\x00', 'NaN', '0x10'])
def test_json_reader_rejects_what_the_grammar_rejects(text):
    with pytest.raises(host.HostError):
        host.json_dump(text)


# ---- PNG -------------------------------------------------------------------------------------------------------
def random_samples(rng, h, w, c, bit_depth):
    return rng.integers(0, 1 << bit_depth, (h, w, c), dtype=np.uint32)


def expected_rgba(samples, colour_type, bit_depth, palette=None, trns=None):
    """What the file means as RGBA8 (16-bit samples keep their high byte, 1/2/4-bit grey scales to the full range)."""
    s = np.asarray(samples, np.uint32)
    h, w, _ = s.shape
    maxv = (1 << bit_depth) - 1
    to8 = (lambda v: v >> 8) if bit_depth == 16 else (lambda v: v) if bit_depth == 8 else (lambda v: v * 255 // maxv)
    out = np.zeros((h, w, 4), np.uint32)
    out[..., 3] = 255
    if colour_type == 0:
        out[..., 0] = out[..., 1] = out[..., 2] = to8(s[..., 0])
        if trns is not None:
            key = struct.unpack(">H", bytes(trns))[0]
            out[..., 3] = np.where(s[..., 0] == key, 0, 255)
    elif colour_type == 2:
        out[..., :3] = to8(s)
        if trns is not None:
            key = struct.unpack(">HHH", bytes(trns))
            out[..., 3] = np.where((s[..., 0] == key[0]) & (s[..., 1] == key[1]) & (s[..., 2] == key[2]), 0, 255)
    elif colour_type == 3:
        pal = np.asarray(palette, np.uint32).reshape(-1, 3)
        out[..., :3] = pal[s[..., 0]]
        if trns is not None:
            a = np.full(pal.shape[0], 255, np.uint32)
            a[:len(trns)] = np.frombuffer(bytes(trns), np.uint8)
            out[..., 3] = a[s[..., 0]]
    elif colour_type == 4:
        out[..., 0] = out[..., 1] = out[..., 2] = to8(s[..., 0])
        out[..., 3] = to8(s[..., 1])
    else:
        out[...] = to8(s)
    return out.astype(np.uint8)


PNG_CASES = [(ct, bd) for ct, bds in ((0, (1, 2, 4, 8, 16)), (2, (8, 16)), (3, (1, 2, 4, 8)), (4, (8, 16)), (6, (8, 16))) for bd in bds]


@pytest.mark.parametrize("colour_type,bit_depth", PNG_CASES)
@pytest.mark.parametrize("interlace", [False, True])
def test_png_every_colour_type_bit_depth_and_filter(colour_type, bit_depth, interlace):
    rng = np.random.default_rng(colour_type * 100 + bit_depth + (7 if interlace else 0))
    c = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[colour_type]
    h, w = 13, 11                                   # odd sizes: partial bytes at low bit depths, short Adam7 passes
    samples = random_samples(rng, h, w, c, bit_depth)
    palette = rng.integers(0, 256, ((1 << bit_depth), 3), dtype=np.uint32) if colour_type == 3 else None
    trns = None
    if colour_type == 3:
        trns = bytes(rng.integers(0, 256, max(1, (1 << bit_depth) // 2), dtype=np.uint8))
    elif colour_type == 0:
        trns = struct.pack(">H", int(samples[0, 0, 0]))
    elif colour_type == 2:
        trns = struct.pack(">HHH", *[int(v) for v in samples[2, 3]])
    data = gltf_io.encode_png(samples, colour_type, bit_depth, filter_type="cycle", interlace=interlace, palette=palette, trns=trns, idat_split=37)
    got, channels = host.image_decode(data)
    assert np.array_equal(got, expected_rgba(samples, colour_type, bit_depth, palette, trns))
    assert channels == {0: 2, 2: 4, 3: 4, 4: 2, 6: 4}[colour_type]     # tRNS adds an alpha channel


@pytest.mark.parametrize("level", [0, 1, 9])
def test_png_inflate_stored_fixed_and_dynamic_blocks(level):
    rng = np.random.default_rng(level)
    # level 0 -> stored blocks; a tiny image -> fixed Huffman codes; a large smooth one -> dynamic codes with long matches
    small = rng.integers(0, 256, (2, 3, 4), dtype=np.uint32)
    y, x = np.mgrid[0:200, 0:300]
    big = np.stack([(x + y) & 255, (x * 3) & 255, (y // 5) & 255, np.full_like(x, 255)], 2).astype(np.uint32)
    big[50:60] = rng.integers(0, 256, (10, 300, 4))
    for samples in (small, big):
        got, _ = host.image_decode(gltf_io.encode_png(samples, 6, 8, filter_type=4 if samples is big else 0, level=level))
        assert np.array_equal(got, samples.astype(np.uint8))


def test_png_matches_pillow_on_a_pillow_written_file():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (64, 48, 4), dtype=np.uint8)
    for mode, arr in (("RGBA", a), ("RGB", a[..., :3]), ("L", a[..., 0]), ("P", None)):
        buf = io.BytesIO()
        im = Image.fromarray(a[..., :3]).quantize(17) if mode == "P" else Image.fromarray(arr, mode)
        im.save(buf, format="PNG", optimize=True)
        got, _ = host.image_decode(buf.getvalue())
        want = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGBA"))
        assert np.array_equal(got, want)


def test_png_writer_round_trip():
    rng = np.random.default_rng(9)
    for h, w in ((1, 1), (7, 5), (300, 257)):          # the last one needs several stored blocks
        rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        data = host.image_encode_png(rgb)
        back, channels = host.image_decode(data)
        assert channels == 3 and np.array_equal(back[..., :3], rgb) and (back[..., 3] == 255).all()
        try:
            from PIL import Image
            assert np.array_equal(np.asarray(Image.open(io.BytesIO(data)).convert("RGB")), rgb)
        except ImportError:
            pass


def test_png_errors():
    good = gltf_io.encode_png(np.zeros((4, 4, 4), np.uint32), 6, 8)
    for bad in (good[:-20], good[:40] + bytes([good[40] ^ 1]) + good[41:], b"\x89PNG\r\n\x1a\n" + b"\0" * 30, b"not an image at all"):
        with pytest.raises(host.HostError):
            host.image_decode(bad)


# ---- JPEG ------------------------------------------------------------------------------------------------------
def smooth_image(h, w, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.stack([127 + 120 * np.sin(x / 9.0 + seed), 127 + 120 * np.cos(y / 7.0), 127 + 100 * np.sin((x + y) / 13.0)], 2)
    img += rng.normal(0, 6, img.shape)
    img[h // 3:h // 3 + 9, w // 4:w // 4 + 17] = [250, 10, 10]      # a hard chroma edge
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("subsampling,quality,size,restart", [(0, 95, (40, 56), 0), (1, 90, (41, 59), 0), (2, 85, (67, 45), 0),
                                                              (2, 92, (64, 64), 3), (0, 75, (8, 8), 0), (2, 90, (1, 1), 0)])
def test_jpeg_matches_pillow_within_decoder_tolerance(subsampling, quality, size, restart):
    """IDCT rounding and chroma upsampling are decoder-defined: compared with libjpeg (through Pillow) by tolerance."""
    Image = pytest.importorskip("PIL.Image")
    img = smooth_image(size[0], size[1], subsampling + quality)
    buf = io.BytesIO()
    kw = dict(restart_marker_blocks=restart) if restart else {}
    Image.fromarray(img).save(buf, format="JPEG", quality=quality, subsampling=subsampling, **kw)
    got, channels = host.image_decode(buf.getvalue())
    want = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB")).astype(np.int32)
    assert channels == 3 and got.shape == (size[0], size[1], 4) and (got[..., 3] == 255).all()
    diff = np.abs(got[..., :3].astype(np.int32) - want)
    # 4:4:4 differs by IDCT / colour-conversion rounding only; subsampled chroma also by the rounding of the upsampling filter
    assert diff.mean() < (0.1 if subsampling == 0 else 0.6), diff.mean()
    assert diff.max() <= 3, diff.max()


REFERENCE_JPEG = "/root/reference/result.jpg"


@pytest.mark.skipif(not os.path.exists(REFERENCE_JPEG), reason="the reference checkout (its only image file) is not on this machine")
def test_jpeg_decodes_the_reference_repositorys_own_picture():
    Image = pytest.importorskip("PIL.Image")
    data = open(REFERENCE_JPEG, "rb").read()
    got, channels = host.image_decode(data)
    want = np.asarray(Image.open(io.BytesIO(data)).convert("RGB")).astype(np.int32)
    assert got.shape == (1080, 1920, 4) and channels == 3
    diff = np.abs(got[..., :3].astype(np.int32) - want)
    assert diff.max() <= 3 and diff.mean() < 0.1


def test_jpeg_greyscale_and_errors():
    Image = pytest.importorskip("PIL.Image")
    img = smooth_image(33, 47, 3)[..., 0]
    buf = io.BytesIO()
    Image.fromarray(img, "L").save(buf, format="JPEG", quality=90)
    got, channels = host.image_decode(buf.getvalue())
    want = np.asarray(Image.open(io.BytesIO(buf.getvalue()))).astype(np.int32)
    assert channels == 1 and np.abs(got[..., 0].astype(np.int32) - want).max() <= 2
    assert np.array_equal(got[..., 0], got[..., 1]) and np.array_equal(got[..., 0], got[..., 2])
    with pytest.raises(host.HostError):
        host.image_decode(buf.getvalue()[:200])


@pytest.mark.parametrize("subsampling,quality,size,mode,restart", [(0, 95, (40, 56), "RGB", 0), (2, 90, (67, 45), "RGB", 0), (1, 80, (41, 59), "RGB", 0),
                                                                   (2, 92, (64, 64), "RGB", 3), (0, 90, (33, 47), "L", 0), (2, 30, (96, 80), "RGB", 0),
                                                                   (0, 100, (24, 24), "RGB", 0)])
def test_progressive_jpeg_matches_pillow_within_decoder_tolerance(subsampling, quality, size, mode, restart):
    """Progressive files (spectral selection + successive approximation, DC and AC refinement scans, end-of-band runs):
    the coefficients must come out exactly as libjpeg's, so the tolerance is the sequential one."""
    Image = pytest.importorskip("PIL.Image")
    img = smooth_image(size[0], size[1], subsampling + quality)
    if mode == "L":
        img = img[..., 0]
    buf = io.BytesIO()
    kw = dict(restart_marker_blocks=restart) if restart else {}
    if mode == "RGB":
        kw["subsampling"] = subsampling
    Image.fromarray(img, mode).save(buf, format="JPEG", quality=quality, progressive=True, **kw)
    data = buf.getvalue()
    assert b"\xff\xc2" in data, "a progressive frame header"
    assert data.count(b"\xff\xda") > 3, "several scans"
    got, channels = host.image_decode(data)
    want = np.asarray(Image.open(io.BytesIO(data)).convert("RGB")).astype(np.int32)
    assert channels == (3 if mode == "RGB" else 1) and got.shape == (size[0], size[1], 4)
    diff = np.abs(got[..., :3].astype(np.int32) - want)
    assert diff.mean() < (0.1 if subsampling == 0 or mode == "L" else 0.6), diff.mean()
    assert diff.max() <= 3, diff.max()
    # and the same picture coded sequentially decodes to the same samples up to the coder's own choice of tables
    seq = io.BytesIO()
    Image.fromarray(img, mode).save(seq, format="JPEG", quality=quality, **kw)
    got_seq, _ = host.image_decode(seq.getvalue())
    assert np.array_equal(got, got_seq), "progressive and sequential codings of the same coefficients decode alike"


def test_progressive_jpeg_cut_after_its_first_scan():
    """Only the first scan (DC of all components) followed by the end marker: a blocky but valid picture, as the format
    intends for partial transmissions.  (The later scans of a progressive file each hold ONE component: the per-component
    scan walk is what the test above exercises.)"""
    Image = pytest.importorskip("PIL.Image")
    img = smooth_image(37, 53, 9)
    prog = io.BytesIO()
    Image.fromarray(img).save(prog, format="JPEG", quality=90, subsampling=2, progressive=True)
    p = prog.getvalue()
    first_scan = p.index(b"\xff\xda")
    second_scan = p.index(b"\xff\xda", first_scan + 2)
    tables = p.rfind(b"\xff\xc4", first_scan, second_scan)     # the Huffman tables sent ahead of the second scan
    cut = p[:tables if tables > first_scan else second_scan] + b"\xff\xd9"
    got, _ = host.image_decode(cut)
    want = np.asarray(Image.open(io.BytesIO(p)).convert("RGB")).astype(np.int32)
    assert got.shape == (37, 53, 4)
    err = np.abs(got[..., :3].astype(np.int32) - want).mean()
    assert 1.0 < err < 40.0, err


# ---- glTF -> flat arrays -----------------------------------------------------------------------------------------
def assert_same_scene(got, want):
    for k in ("indices", "positions", "normals", "uvs"):
        assert np.array_equal(np.asarray(got[k]).view(np.uint32), np.asarray(want[k]).view(np.uint32)), k
    gm, wm = got["materials"], want["materials"]
    for f in ("base_color", "albedo_tex", "mr_tex", "metalness", "roughness"):
        assert np.array_equal(gm[f], wm[f]), f
    assert len(got["textures"]) == len(want["textures"])
    for a, b in zip(got["textures"][1:], want["textures"][1:]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("kw", [dict(), dict(container="glb"), dict(embed=True), dict(interleave=True, primitives_per_mesh=1),
                                dict(index_type="u32", primitives_per_mesh=3)])
def test_gltf_round_trip_reproduces_the_flat_arrays_bit_for_bit(tmp_path, kw):
    sc = scenes.generate("atrium", 6000, texture_size=16)
    assert sc["primitive_count"] > 8                       # several meshes, a node hierarchy of depth > 2
    path = gltf_io.save_gltf(sc, str(tmp_path), "round trip", **kw)
    assert_same_scene(host.gltf_load(path), sc)


def test_gltf_tiny_scene_with_untextured_material_and_u8_indices(tmp_path):
    sc = scenes.tiny_scene()
    path = gltf_io.save_gltf(sc, str(tmp_path), "tiny", index_type="auto", primitives_per_mesh=2)
    got = host.gltf_load(path)
    assert_same_scene(got, sc)
    assert (got["materials"]["albedo_tex"] == 0).any()     # no baseColorTexture -> texture 0 (opaque white in the kernels)


def test_gltf_wide_index_layout(tmp_path):
    """wide_indices=1: four plain words per triangle (the layout for scenes past 2^24 vertices) naming the same triangles."""
    sc = scenes.generate("atrium", 3000, texture_size=8)
    path = gltf_io.save_gltf(sc, str(tmp_path), "wide")
    packed, wide = host.gltf_load(path), host.gltf_load(path, wide_indices=1)
    assert packed["index_words"] == 3 and wide["index_words"] == 4
    want = scenes.widen_indices(sc)
    assert np.array_equal(wide["indices"], want["indices"])
    assert np.array_equal(wide["triangles"], packed["triangles"])
    p = str(tmp_path / "wide.ptrscene")
    scenes.save_ptrscene(want, p)
    back = host.ptrscene_load(p)
    assert back["index_words"] == 4 and np.array_equal(back["indices"], want["indices"])


def test_ptrscene_file_round_trip(tmp_path):
    sc = scenes.tiny_scene()
    p = str(tmp_path / "tiny.ptrscene")
    scenes.save_ptrscene(sc, p)
    assert_same_scene(host.ptrscene_load(p), sc)


def write_handmade(tmp_path, doc, blob):
    doc = dict(doc)
    doc.setdefault("asset", {"version": "2.0"})
    doc["buffers"] = [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}]
    p = tmp_path / "hand.gltf"
    p.write_text(json.dumps(doc))
    return str(p)


def test_gltf_format_corner_cases(tmp_path):
    """Accessor byteOffset, normalised integer uvs, no indices, no normals (computed), no material (format default),
    the same mesh under two nodes, node transforms ignored or applied."""
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], "<f4")
    uv16 = np.array([[0, 0], [65535, 0], [0, 65535], [65535, 0], [65535, 65535], [0, 32768]], "<u2")
    blob = b"\xAA" * 8 + pos.tobytes() + uv16.tobytes()
    doc = {
        "scenes": [{"nodes": [0, 1]}],
        "nodes": [{"mesh": 0, "translation": [10, 0, 0], "children": [2]}, {"mesh": 0, "scale": [2, 2, 2]},
                  {"mesh": 0, "rotation": [0, 0, 0.7071067811865476, 0.7071067811865476]}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "TEXCOORD_0": 1}}]}],
        "accessors": [{"bufferView": 0, "byteOffset": 8, "componentType": 5126, "count": 6, "type": "VEC3"},
                      {"bufferView": 1, "componentType": 5123, "normalized": True, "count": 6, "type": "VEC2"}],
        "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 8 + pos.nbytes}, {"buffer": 0, "byteOffset": 8 + pos.nbytes, "byteLength": uv16.nbytes}],
    }
    path = write_handmade(tmp_path, doc, blob)
    plain = host.gltf_load(path)
    # breadth first: node0, node1, then node0's child; transforms ignored -> three copies of the same two triangles
    assert plain["positions"].size == 3 * 18 and np.array_equal(plain["positions"].reshape(3, 6, 3), np.stack([pos] * 3))
    assert np.array_equal(plain["indices"] & 0xFFFFFF, np.arange(18))
    assert [(int(a) >> 24, int(b) >> 24, int(c) >> 24) for a, b, c in plain["indices"].reshape(-1, 3)[::2]] == [(0, 0, 0), (0, 0, 1), (0, 0, 2)]
    assert np.array_equal(plain["uvs"][:12], (uv16.astype(np.float64) / 65535.0).astype(np.float32).reshape(-1))
    assert np.array_equal(plain["normals"].reshape(-1, 3), np.tile([0, 0, 1], (18, 1)).astype(np.float32))   # computed, +z
    m = plain["materials"]
    assert len(m) == 3 and (m["albedo_tex"] == 0).all() and (m["metalness"] == 1).all() and (m["roughness"] == 1).all()
    assert np.array_equal(m["base_color"], np.ones((3, 4), np.float32))

    moved = host.gltf_load(path, apply_transforms=True)
    P = moved["positions"].reshape(3, 6, 3)
    assert np.array_equal(P[0], pos + np.array([10, 0, 0], np.float32))
    assert np.array_equal(P[1], pos * 2)
    # child of node0: translate(10,0,0) * rotate 90 degrees about z: (x, y) -> (-y, x)
    want = np.stack([-pos[:, 1] + 10, pos[:, 0], pos[:, 2]], 1)
    assert np.allclose(P[2], want, atol=1e-6)
    assert np.allclose(moved["normals"].reshape(-1, 3), np.tile([0, 0, 1], (18, 1)), atol=1e-7)


def test_gltf_errors(tmp_path):
    pos = np.zeros((3, 3), "<f4")
    base = {"scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0}], "meshes": [{"primitives": [{"attributes": {"POSITION": 0}}]}],
            "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"}],
            "bufferViews": [{"buffer": 0, "byteLength": pos.nbytes}]}
    assert host.gltf_load(write_handmade(tmp_path, base, pos.tobytes()))["indices"].size == 3

    def broken(**edits):
        d = json.loads(json.dumps(base))
        for k, v in edits.items():
            d[k] = v
        return write_handmade(tmp_path, d, pos.tobytes())
    for path in (broken(accessors=[{"bufferView": 0, "componentType": 5126, "count": 4, "type": "VEC3"}]),      # runs past its view
                 broken(accessors=[{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC2"}]),      # wrong shape
                 broken(meshes=[{"primitives": [{"attributes": {"POSITION": 0}, "mode": 1}]}]),                 # lines
                 broken(meshes=[{"primitives": [{"attributes": {"POSITION": 0}, "material": 3}]}]),             # no such material
                 broken(nodes=[{"mesh": 5}]),
                 broken(scenes=[]),
                 str(tmp_path / "missing.gltf")):
        with pytest.raises(host.HostError):
            host.gltf_load(path)
    (tmp_path / "garbage.gltf").write_text("{ not json")
    with pytest.raises(host.HostError):
        host.gltf_load(str(tmp_path / "garbage.gltf"))


def test_instanced_flattening_keeps_every_mesh_once(tmp_path):
    """gltf::flatten_instanced: meshes shared between nodes come out once, untransformed, each an object of its own, and
    every node becomes a placement with its world matrix."""
    from common import make_bundle
    b = make_bundle("atrium", 3000, 32, 24)
    n_meshes = (len(b["materials"]) + 3) // 4
    rng = np.random.default_rng(3)
    mats = []
    for copy in range(3):
        m = np.zeros((3, 4), np.float32)
        m[:, :3] = np.eye(3) * (1.0 + 0.25 * copy)
        m[0, 1] = 0.125 * copy
        m[:, 3] = rng.uniform(-20, 20, 3).astype(np.float32)
        mats.append(m)
    placements = [(mesh, mats[copy]) for copy in range(3) for mesh in range(n_meshes)]
    path = gltf_io.save_gltf(b, str(tmp_path), "placed", placements=placements)
    sc = host.gltf_load(path, instanced=True)
    for key in ("positions", "normals", "uvs"):
        assert np.array_equal(sc[key], b[key]), key
    assert np.array_equal(sc["indices"], b["indices"])
    assert len(sc["object_first"]) == n_meshes + 1 and sc["object_first"][0] == 0 and sc["object_first"][-1] == len(b["indices"]) // 3
    ranges = gltf_io.primitive_ranges(b)
    for k in range(n_meshes):
        assert sc["object_first"][k] == ranges[4 * k][0]
    assert np.array_equal(sc["object_of_copy"], np.array([mesh for _, mesh in [(c, m) for c in range(3) for m in range(n_meshes)]], np.uint32))
    assert np.array_equal(sc["object_to_world"], np.stack([mats[copy] for copy in range(3) for _ in range(n_meshes)]))
    # the plain flattening of the same file writes every node's mesh out
    plain = host.gltf_load(path, apply_transforms=True)
    assert len(plain["indices"]) == 3 * len(b["indices"])
