"""Writers for the files the scene loader reads: glTF 2.0 (.gltf + .bin + images, or one .glb) from the flat scene
arrays of `scenes.generate`, and a PNG encoder that can produce every PNG flavour (colour type, bit depth, filter,
Adam7) so that the C++ decoder is exercised on more than one file layout.  Test and benchmark tooling only: the
render path never writes these files."""
import base64
import json
import os
import struct
import zlib

import numpy as np

# ---- PNG -----------------------------------------------------------------------------------------------------
_ADAM7 = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]  # x0, y0, dx, dy


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def _filter_rows(rows, bpp, filter_type):
    """rows: (h, stride) uint8 of packed scanlines.  filter_type 0..4, or 'cycle' (row y uses y % 5)."""
    h, stride = rows.shape
    out = bytearray()
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        cur = rows[y].astype(np.int32)
        ft = (y % 5) if filter_type == "cycle" else filter_type
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]]) if stride > bpp else np.zeros(stride, np.int32)
        up_left = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]]) if stride > bpp else np.zeros(stride, np.int32)
        if ft == 0:
            f = cur
        elif ft == 1:
            f = cur - left
        elif ft == 2:
            f = cur - prev
        elif ft == 3:
            f = cur - ((left + prev) >> 1)
        else:
            pred = np.array([_paeth(int(a), int(b), int(c)) for a, b, c in zip(left, prev, up_left)], np.int32)
            f = cur - pred
        out.append(ft)
        out += (f & 0xFF).astype(np.uint8).tobytes()
        prev = cur
    return bytes(out)


def _chunk(kind, data):
    return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)


def encode_png(samples, colour_type, bit_depth=8, filter_type=0, interlace=False, palette=None, trns=None, idat_split=None, level=6):
    """samples: (h, w, c) integer array of samples at `bit_depth` (c = 1 grey / index, 2 grey+alpha, 3 RGB, 4 RGBA).
    colour_type: 0 grey, 2 RGB, 3 indexed, 4 grey+alpha, 6 RGBA.  Returns the file's bytes."""
    samples = np.asarray(samples)
    h, w, c = samples.shape
    assert c == {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[colour_type]

    def pack(sub):
        hh, ww, _ = sub.shape
        if bit_depth == 8:
            return sub.astype(np.uint8).reshape(hh, ww * c)
        if bit_depth == 16:
            return sub.astype(">u2").view(np.uint8).reshape(hh, ww * c * 2)
        per = 8 // bit_depth                                   # 1, 2, 4 bits: c == 1, packed MSB first
        pad = (-ww) % per
        v = np.concatenate([sub[:, :, 0], np.zeros((hh, pad), sub.dtype)], 1).astype(np.uint16).reshape(hh, -1, per)
        shifts = np.array([8 - bit_depth * (i + 1) for i in range(per)], np.uint16)
        return (v << shifts).sum(axis=2).astype(np.uint8)

    bpp = max(1, c * bit_depth // 8)
    raw = b""
    if not interlace:
        raw = _filter_rows(pack(samples), bpp, filter_type)
    else:
        for x0, y0, dx, dy in _ADAM7:
            sub = samples[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                raw += _filter_rows(pack(sub), bpp, filter_type)
    comp = zlib.compress(raw, level)
    out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bit_depth, colour_type, 0, 0, 1 if interlace else 0))
    if palette is not None:
        out += _chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    if trns is not None:
        out += _chunk(b"tRNS", bytes(trns))
    out += _chunk(b"tEXt", b"Comment\x00ancillary chunks are skipped")
    parts = [comp] if not idat_split else [comp[i:i + idat_split] for i in range(0, len(comp), idat_split)]
    for p in parts:
        out += _chunk(b"IDAT", p)
    return out + _chunk(b"IEND", b"")


def encode_png_rgba8(texels, filter_type="cycle"):
    return encode_png(np.asarray(texels, np.uint8), 6, 8, filter_type)


# ---- glTF ----------------------------------------------------------------------------------------------------
def primitive_ranges(scene):
    """Splits the flat arrays back into the primitives they were emitted as: one material id per primitive (packed on
    top of the index words, src/scene.cpp:140-142), vertices of a primitive contiguous.
    Returns a list of (first_triangle, triangle_count, first_vertex, vertex_count)."""
    idx = np.asarray(scene["indices"], np.uint32).reshape(-1, 3)
    mat = ((idx[:, 0] >> 24) << 16) | ((idx[:, 1] >> 24) << 8) | (idx[:, 2] >> 24)
    vid = idx & np.uint32(0x00FFFFFF)
    cuts = np.concatenate([[0], np.nonzero(np.diff(mat))[0] + 1, [idx.shape[0]]])
    out = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        lo, hi = int(vid[a:b].min()), int(vid[a:b].max())
        out.append((int(a), int(b - a), lo, hi - lo + 1))
    return out


def _json_float(x):
    return float(np.float32(x))  # repr of the double that equals the float32: survives text -> double -> float exactly


def save_gltf(scene, directory, name="scene", primitives_per_mesh=4, container="gltf", image_format="png", embed=False,
              index_type="auto", interleave=False, placements=None):
    """Writes `scene` as <directory>/<name>.gltf (+ .bin + one image file per texture), or <name>.glb.
    The node tree is a binary heap (children of node i are 2i+1, 2i+2), so a breadth-first walk meets the meshes in
    file order and the loader's flat arrays come out identical to `scene`'s.
    interleave: positions/normals/uvs of a primitive share one strided buffer view (byteStride 32).
    placements: list of (mesh number, (3, 4) object-to-world matrix) -- the node list then holds one node per placement,
    in that order, instead of one node per mesh (meshes shared between nodes: what an instancing loader looks for)."""
    os.makedirs(directory, exist_ok=True)
    positions = np.asarray(scene["positions"], np.float32).reshape(-1, 3)
    normals = np.asarray(scene["normals"], np.float32).reshape(-1, 3)
    uvs = np.asarray(scene["uvs"], np.float32).reshape(-1, 2)
    idx = (np.asarray(scene["indices"], np.uint32) & np.uint32(0x00FFFFFF)).reshape(-1, 3)
    prims = primitive_ranges(scene)
    assert len(prims) == len(scene["materials"])

    blob = bytearray()
    views, accessors = [], []

    def add_view(data, stride=None, target=None):
        while len(blob) % 4:
            blob.append(0)
        v = {"buffer": 0, "byteOffset": len(blob), "byteLength": len(data)}
        if stride:
            v["byteStride"] = stride
        if target:
            v["target"] = target
        blob.extend(data)
        views.append(v)
        return len(views) - 1

    def add_accessor(view, component_type, count, kind, offset=0, minmax=None):
        a = {"bufferView": view, "componentType": component_type, "count": count, "type": kind}
        if offset:
            a["byteOffset"] = offset
        if minmax is not None:
            a["min"] = [_json_float(x) for x in minmax[0]]
            a["max"] = [_json_float(x) for x in minmax[1]]
        accessors.append(a)
        return len(accessors) - 1

    gl_prims = []
    for p, (t0, tn, v0, vn) in enumerate(prims):
        local = (idx[t0:t0 + tn] - v0).reshape(-1)
        kind = index_type if index_type != "auto" else ("u16" if vn <= 65535 else "u32")
        if kind == "u8":
            assert vn <= 255
            ia = add_accessor(add_view(local.astype(np.uint8).tobytes(), target=34963), 5121, local.size, "SCALAR")
        elif kind == "u16":
            assert vn <= 65535
            ia = add_accessor(add_view(local.astype("<u2").tobytes(), target=34963), 5123, local.size, "SCALAR")
        else:
            ia = add_accessor(add_view(local.astype("<u4").tobytes(), target=34963), 5125, local.size, "SCALAR")
        P, N, U = positions[v0:v0 + vn], normals[v0:v0 + vn], uvs[v0:v0 + vn]
        mm = (P.min(axis=0), P.max(axis=0))
        if interleave:
            view = add_view(np.concatenate([P, N, U], 1).astype("<f4").tobytes(), stride=32, target=34962)
            pa = add_accessor(view, 5126, vn, "VEC3", 0, mm)
            na = add_accessor(view, 5126, vn, "VEC3", 12)
            ua = add_accessor(view, 5126, vn, "VEC2", 24)
        else:
            pa = add_accessor(add_view(P.astype("<f4").tobytes(), target=34962), 5126, vn, "VEC3", 0, mm)
            na = add_accessor(add_view(N.astype("<f4").tobytes(), target=34962), 5126, vn, "VEC3")
            ua = add_accessor(add_view(U.astype("<f4").tobytes(), target=34962), 5126, vn, "VEC2")
        gl_prims.append({"attributes": {"POSITION": pa, "NORMAL": na, "TEXCOORD_0": ua}, "indices": ia, "material": p, "mode": 4})

    meshes = [{"name": f"mesh{m}", "primitives": gl_prims[i:i + primitives_per_mesh]}
              for m, i in enumerate(range(0, len(gl_prims), primitives_per_mesh))]
    n_nodes = len(meshes)
    nodes = []
    for i in range(n_nodes):
        nd = {"name": f"node{i}", "mesh": i}
        kids = [k for k in (2 * i + 1, 2 * i + 2) if k < n_nodes]
        if kids:
            nd["children"] = kids
        nodes.append(nd)

    if placements is not None:
        nodes = []
        for k, (mesh_number, m) in enumerate(placements):
            m = np.asarray(m, np.float64).reshape(3, 4)
            column_major = [m[r, c] for c in range(4) for r in range(3)]
            column_major = column_major[0:3] + [0.0] + column_major[3:6] + [0.0] + column_major[6:9] + [0.0] + column_major[9:12] + [1.0]
            nodes.append({"name": f"placement{k}", "mesh": int(mesh_number), "matrix": [_json_float(x) for x in column_major]})
        n_nodes = len(nodes)
    scene_nodes = list(range(n_nodes)) if placements is not None else ([0] if n_nodes else [])

    textures = [t for t in scene["textures"] if t is not None]
    images = []
    for i, t in enumerate(textures):
        if image_format == "png":
            data, mime, ext = encode_png_rgba8(t, filter_type=i % 5), "image/png", "png"
        else:
            import io
            from PIL import Image
            buf = io.BytesIO()
            Image.fromarray(np.asarray(t, np.uint8)[:, :, :3]).save(buf, format="JPEG", quality=92)
            data, mime, ext = buf.getvalue(), "image/jpeg", "jpg"
        if container == "glb":
            images.append({"bufferView": add_view(data), "mimeType": mime})
        elif embed:
            images.append({"uri": f"data:{mime};base64," + base64.b64encode(data).decode()})
        else:
            fn = f"{name} tex{i}.{ext}"              # the space exercises percent-decoding of the uri
            with open(os.path.join(directory, fn), "wb") as f:
                f.write(data)
            images.append({"uri": fn.replace(" ", "%20")})

    mats = []
    for m in scene["materials"]:
        pbr = {"baseColorFactor": [_json_float(x) for x in m["base_color"]],
               "metallicFactor": _json_float(m["metalness"]), "roughnessFactor": _json_float(m["roughness"])}
        if int(m["albedo_tex"]):
            pbr["baseColorTexture"] = {"index": int(m["albedo_tex"]) - 1}
        if int(m["mr_tex"]):
            pbr["metallicRoughnessTexture"] = {"index": int(m["mr_tex"]) - 1}
        mats.append({"pbrMetallicRoughness": pbr})

    doc = {"asset": {"version": "2.0", "generator": "path_tracer_amd.gltf_io"}, "scene": 0,
           "scenes": [{"nodes": scene_nodes}], "nodes": nodes, "meshes": meshes, "materials": mats,
           "textures": [{"source": i, "sampler": 0} for i in range(len(textures))], "images": images,
           "samplers": [{"magFilter": 9729, "minFilter": 9729, "wrapS": 10497, "wrapT": 10497}],
           "accessors": accessors, "bufferViews": views}
    if container == "glb":
        doc["buffers"] = [{"byteLength": len(blob)}]
        js = json.dumps(doc, separators=(",", ":")).encode()
        js += b" " * ((-len(js)) % 4)
        binary = bytes(blob) + b"\x00" * ((-len(blob)) % 4)
        path = os.path.join(directory, name + ".glb")
        with open(path, "wb") as f:
            f.write(struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(js) + 8 + len(binary)))
            f.write(struct.pack("<II", len(js), 0x4E4F534A) + js)
            f.write(struct.pack("<II", len(binary), 0x004E4942) + binary)
        return path
    if embed:
        doc["buffers"] = [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(bytes(blob)).decode()}]
    else:
        with open(os.path.join(directory, name + ".bin"), "wb") as f:
            f.write(bytes(blob))
        doc["buffers"] = [{"byteLength": len(blob), "uri": name + ".bin"}]
    path = os.path.join(directory, name + ".gltf")
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)
    return path
