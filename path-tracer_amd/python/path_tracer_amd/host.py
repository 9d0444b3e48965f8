"""ctypes binding of host/libptr_host.so: the scene-preparation code (BVH builder + GPU flattener, camera)."""
import ctypes as C
import os

import numpy as np

from . import paths

_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(paths.HOST_LIB)
        L.ptrh_last_error.restype = C.c_char_p
        L.ptrh_bvh_build.restype = C.c_int64
        L.ptrh_bvh_build.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
        L.ptrh_bvh_visit_ranks.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.ptrh_bvh_flatten.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_uint64)]
        L.ptrh_bvh_flatten_spheres.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_uint32, C.POINTER(C.c_void_p),
                                               C.POINTER(C.c_uint64)]
        L.ptrh_bvh_flatten_spheres.restype = C.c_int32
        L.ptrh_random_spheres.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.ptrh_random_spheres.restype = C.c_int32
        L.ptrh_bvh_flatten_instances.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int64, C.c_uint32, C.POINTER(C.c_void_p),
                                                 C.POINTER(C.c_uint64)]
        L.ptrh_bvh_flatten_instances.restype = C.c_int32
        L.ptrh_bvh_flatten_objects.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_uint32, C.POINTER(C.c_void_p),
                                               C.POINTER(C.c_uint64), C.c_void_p]
        L.ptrh_bvh_flatten_objects.restype = C.c_int32
        L.ptrh_bvh_flatten_placements.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_uint32, C.POINTER(C.c_void_p),
                                                  C.POINTER(C.c_uint64)]
        L.ptrh_bvh_flatten_placements.restype = C.c_int32
        L.ptrh_free.argtypes = [C.c_void_p]
        L.ptrh_camera.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p,
                                  C.c_int32, C.c_void_p]
        vp = C.c_void_p
        L.ptrh_gltf_load.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(vp)]
        L.ptrh_scene_load.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.ptrh_scene_counts.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.ptrh_scene_arrays.argtypes = [vp, vp, vp, vp, vp, vp]
        L.ptrh_scene_texture.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), vp]
        L.ptrh_scene_placements.argtypes = [vp, C.POINTER(C.c_uint64), vp, vp, vp]
        L.ptrh_scene_free.argtypes = [vp]
        L.ptrh_image_decode.argtypes = [vp, C.c_uint64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(vp)]
        L.ptrh_image_encode_png.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(vp), C.POINTER(C.c_uint64)]
        L.ptrh_json_dump.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(vp)]
        _lib = L
    return _lib


class HostError(RuntimeError):
    pass


REFERENCE_ORDER = 0
SAH_PARTITIONED = 1


def bvh_build(triangles, mode=REFERENCE_ORDER):
    """triangles: (N, 9) float32.  Returns the threaded node array as an (2N-1, 8) uint32 view
    (min.xyz, next_id, max.xyz, primitive_id per 32-byte record)."""
    tri = np.ascontiguousarray(triangles, dtype=np.float32).reshape(-1, 9)
    n = tri.shape[0]
    nodes = np.zeros((max(1, 2 * n - 1), 8), dtype=np.uint32)
    got = lib().ptrh_bvh_build(tri.ctypes.data, n, mode, nodes.ctypes.data)
    if got < 0:
        raise HostError(lib().ptrh_last_error().decode())
    return nodes[:got]


def bvh_visit_ranks(nodes, primitive_count):
    nodes = np.ascontiguousarray(nodes)
    ranks = np.zeros(primitive_count, dtype=np.int32)
    if lib().ptrh_bvh_visit_ranks(nodes.ctypes.data, nodes.shape[0], primitive_count, ranks.ctypes.data) != 0:
        raise HostError(lib().ptrh_last_error().decode())
    return ranks


def bvh_flatten(triangles, ranks=None, threads=0, spheres=None):
    """Returns the GPU layout blob (uint8 array): 64-byte header | 64-byte nodes | 64-byte triangles.
    spheres: optional (S, 4) array of centre + radius, the sphere leaves of the same tree."""
    tri = np.ascontiguousarray(triangles, dtype=np.float32).reshape(-1, 9)
    n = tri.shape[0]
    r = None if ranks is None else np.ascontiguousarray(ranks, dtype=np.int32)
    ptr = C.c_void_p()
    size = C.c_uint64()
    if spheres is not None and len(spheres):
        sp = np.ascontiguousarray(spheres, dtype=np.float32).reshape(-1, 4)
        rc = lib().ptrh_bvh_flatten_spheres(tri.ctypes.data if n else None, n, None if r is None else r.ctypes.data, sp.ctypes.data, sp.shape[0], threads,
                                            C.byref(ptr), C.byref(size))
    else:
        rc = lib().ptrh_bvh_flatten(tri.ctypes.data, n, None if r is None else r.ctypes.data, threads, C.byref(ptr), C.byref(size))
    if rc != 0:
        raise HostError(lib().ptrh_last_error().decode())
    try:
        blob = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(size.value,)).copy()
    finally:
        lib().ptrh_free(ptr)
    return blob


def random_spheres(seed):
    """The reference's old scene of spheres (its commented-out scene::random_scene, src/scene.cpp:11-58), deterministic in `seed`:
    an (n, 9) float32 array of x y z radius r g b metalness roughness."""
    count = C.c_uint32()
    if lib().ptrh_random_spheres(seed, None, 0, C.byref(count)) != 0:
        raise HostError(lib().ptrh_last_error().decode())
    out = np.zeros((count.value, 9), np.float32)
    if lib().ptrh_random_spheres(seed, out.ctypes.data, count.value, C.byref(count)) != 0:
        raise HostError(lib().ptrh_last_error().decode())
    return out


def bvh_flatten_instances(object_blob, object_to_world, threads=0):
    """Top level over placed copies of one flattened tree.  object_to_world: (n, 3, 4) affine matrices.
    Returns the blob (uint8 array): 64-byte header | 64-byte nodes | 64-byte instance records."""
    obj = np.ascontiguousarray(object_blob, dtype=np.uint8)
    m = np.ascontiguousarray(object_to_world, dtype=np.float32).reshape(-1, 12)
    ptr = C.c_void_p()
    size = C.c_uint64()
    rc = lib().ptrh_bvh_flatten_instances(obj.ctypes.data, obj.size, m.ctypes.data, m.shape[0], threads, C.byref(ptr), C.byref(size))
    if rc != 0:
        raise HostError(lib().ptrh_last_error().decode())
    try:
        blob = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(size.value,)).copy()
    finally:
        lib().ptrh_free(ptr)
    return blob


OBJECT_DTYPE = np.dtype([("root", np.int32), ("max_depth", np.uint32), ("box_min", np.float32, 3), ("box_max", np.float32, 3)])


def bvh_flatten_objects(triangles, object_first, ranks=None, threads=0):
    """Several objects in one flattened layout: triangles [object_first[k], object_first[k + 1]) form object k.
    Returns (blob, objects): objects is an OBJECT_DTYPE array, one entry per object."""
    tri = np.ascontiguousarray(triangles, dtype=np.float32).reshape(-1, 9)
    first = np.ascontiguousarray(object_first, dtype=np.uint32)
    r = None if ranks is None else np.ascontiguousarray(ranks, dtype=np.int32)
    objects = np.zeros(max(1, first.size - 1), OBJECT_DTYPE)
    ptr = C.c_void_p()
    size = C.c_uint64()
    rc = lib().ptrh_bvh_flatten_objects(tri.ctypes.data, tri.shape[0], None if r is None else r.ctypes.data, first.ctypes.data, first.size - 1,
                                        threads, C.byref(ptr), C.byref(size), objects.ctypes.data)
    if rc != 0:
        raise HostError(lib().ptrh_last_error().decode())
    try:
        blob = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(size.value,)).copy()
    finally:
        lib().ptrh_free(ptr)
    return blob, objects


def bvh_flatten_placements(objects, object_of_copy, object_to_world, threads=0):
    """Top level over placed copies of the objects of bvh_flatten_objects: copy i shows objects[object_of_copy[i]]."""
    table = np.ascontiguousarray(objects, dtype=OBJECT_DTYPE)
    which = np.ascontiguousarray(object_of_copy, dtype=np.uint32)
    m = np.ascontiguousarray(object_to_world, dtype=np.float32).reshape(-1, 12)
    ptr = C.c_void_p()
    size = C.c_uint64()
    rc = lib().ptrh_bvh_flatten_placements(table.ctypes.data, table.size, which.ctypes.data, m.ctypes.data, m.shape[0], threads, C.byref(ptr), C.byref(size))
    if rc != 0:
        raise HostError(lib().ptrh_last_error().decode())
    try:
        blob = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(size.value,)).copy()
    finally:
        lib().ptrh_free(ptr)
    return blob


def flat_header(blob):
    u = blob[:64].view(np.uint32)
    q = blob[:64].view(np.uint64)
    return dict(magic=int(u[0]), version=int(u[1]), node_count=int(u[2]), triangle_count=int(u[3]),
                nodes_offset=int(q[2]), triangles_offset=int(q[3]), root=int(blob[:64].view(np.int32)[8]),
                max_depth=int(u[9]), scene_min=blob[:64].view(np.float32)[10:13].copy(),
                scene_max=blob[:64].view(np.float32)[13:16].copy())


def camera(pos, target, v_fov, aspect, aperture, focus, ops=()):
    """Returns the 128-byte camera block (uint8 array).  ops: (opcode, a, b, c) with 0 = set_aspect_ratio(a),
    1 = move((a, b, c)), 2 = rotate_y(a)."""
    pos = np.asarray(pos, np.float32)
    tgt = np.asarray(target, np.float32)
    o = np.asarray(ops, np.float32).reshape(-1, 4)
    out = np.zeros(128, np.uint8)
    lib().ptrh_camera(pos.ctypes.data, tgt.ctypes.data, v_fov, aspect, aperture, focus, o.ctypes.data, o.shape[0],
                      out.ctypes.data)
    return out


HOST_SYMBOLS = ["ptrh_last_error", "ptrh_sizes", "ptrh_bvh_build", "ptrh_bvh_visit_ranks", "ptrh_bvh_flatten", "ptrh_bvh_flatten_instances", "ptrh_bvh_flatten_objects", "ptrh_bvh_flatten_placements", "ptrh_free", "ptrh_camera",
                "ptrh_gltf_load", "ptrh_scene_load", "ptrh_scene_counts", "ptrh_scene_arrays", "ptrh_scene_texture", "ptrh_scene_placements", "ptrh_scene_free",
                "ptrh_image_decode", "ptrh_image_encode_png", "ptrh_json_dump"]

# 32-byte material record of a flat scene before upload (texture numbers are 1-based, 0 = none)
SOURCE_MATERIAL_DTYPE = np.dtype([("base_color", np.float32, 4), ("albedo_tex", np.uint32), ("mr_tex", np.uint32),
                                  ("metalness", np.float32), ("roughness", np.float32)])


def _scene_to_dict(handle):
    L = lib()
    try:
        counts = (C.c_uint64 * 5)()
        if L.ptrh_scene_counts(handle, counts) != 0:
            raise HostError(L.ptrh_last_error().decode())
        n_idx, n_vtx, n_mat, n_tex, index_words = (int(c) for c in counts)
        out = dict(indices=np.zeros(n_idx, np.uint32), positions=np.zeros(n_vtx * 3, np.float32), normals=np.zeros(n_vtx * 3, np.float32),
                   uvs=np.zeros(n_vtx * 2, np.float32))
        mats = np.zeros(n_mat, SOURCE_MATERIAL_DTYPE)
        if L.ptrh_scene_arrays(handle, out["indices"].ctypes.data, out["positions"].ctypes.data, out["normals"].ctypes.data,
                               out["uvs"].ctypes.data, mats.ctypes.data) != 0:
            raise HostError(L.ptrh_last_error().decode())
        # the 48-byte device record layout the rest of the Python tooling uses (sampler 1 wherever a texture is named)
        from .scenes import MATERIAL_DTYPE
        full = np.zeros(n_mat, MATERIAL_DTYPE)
        full["base_color"] = mats["base_color"]
        full["albedo_tex"] = mats["albedo_tex"]; full["albedo_sampler"] = (mats["albedo_tex"] != 0).astype(np.uint32)
        full["mr_tex"] = mats["mr_tex"]; full["mr_sampler"] = (mats["mr_tex"] != 0).astype(np.uint32)
        full["metalness"] = mats["metalness"]; full["roughness"] = mats["roughness"]
        out["materials"] = full
        textures = [None]
        for i in range(n_tex):
            w, h = C.c_uint32(), C.c_uint32()
            L.ptrh_scene_texture(handle, i, C.byref(w), C.byref(h), None)
            t = np.zeros((h.value, w.value, 4), np.uint8)
            if L.ptrh_scene_texture(handle, i, None, None, t.ctypes.data) != 0:
                raise HostError(L.ptrh_last_error().decode())
            textures.append(t)
        out["textures"] = textures
        out["index_words"] = index_words
        pc = (C.c_uint64 * 2)()
        if L.ptrh_scene_placements(handle, pc, None, None, None) != 0:
            raise HostError(L.ptrh_last_error().decode())
        if pc[1]:
            out["object_first"] = np.zeros(int(pc[0]) + 1, np.uint32)
            out["object_of_copy"] = np.zeros(int(pc[1]), np.uint32)
            out["object_to_world"] = np.zeros((int(pc[1]), 3, 4), np.float32)
            L.ptrh_scene_placements(handle, pc, out["object_first"].ctypes.data, out["object_of_copy"].ctypes.data, out["object_to_world"].ctypes.data)
        vid = (out["indices"] & np.uint32(0x00FFFFFF)) if index_words == 3 else out["indices"].reshape(-1, 4)[:, :3].reshape(-1)
        out["triangles"] = np.ascontiguousarray(out["positions"].reshape(-1, 3)[vid].reshape(-1, 9))
        out["primitive_count"] = n_mat
        return out
    finally:
        L.ptrh_scene_free(handle)


def gltf_load(path, apply_transforms=False, wide_indices=-1, instanced=False):
    """gltf(path).flatten(): the flat scene arrays (same dict layout as scenes.generate, without a camera).
    instanced: gltf(path).flatten_instanced() -- every mesh once plus object_first / object_of_copy / object_to_world."""
    h = C.c_void_p()
    if lib().ptrh_gltf_load(os.fsencode(path), 2 if instanced else (1 if apply_transforms else 0), wide_indices, C.byref(h)) != 0:
        raise HostError(lib().ptrh_last_error().decode())
    return _scene_to_dict(h)


def ptrscene_load(path):
    h = C.c_void_p()
    if lib().ptrh_scene_load(os.fsencode(path), C.byref(h)) != 0:
        raise HostError(lib().ptrh_last_error().decode())
    return _scene_to_dict(h)


def image_decode(data):
    """PNG / JPEG bytes -> ((h, w, 4) uint8 RGBA, channels stored in the file)."""
    buf = np.frombuffer(bytes(data), np.uint8)
    w, h, ch = C.c_int32(), C.c_int32(), C.c_int32()
    out = C.c_void_p()
    if lib().ptrh_image_decode(buf.ctypes.data, buf.size, C.byref(w), C.byref(h), C.byref(ch), C.byref(out)) != 0:
        raise HostError(lib().ptrh_last_error().decode())
    try:
        img = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(h.value, w.value, 4)).copy()
    finally:
        lib().ptrh_free(out)
    return img, ch.value


def image_encode_png(rgb):
    """(h, w, 3) uint8 -> PNG file bytes."""
    a = np.ascontiguousarray(rgb, dtype=np.uint8)
    out, size = C.c_void_p(), C.c_uint64()
    if lib().ptrh_image_encode_png(a.ctypes.data, a.shape[1], a.shape[0], C.byref(out), C.byref(size)) != 0:
        raise HostError(lib().ptrh_last_error().decode())
    try:
        return C.string_at(out, size.value)
    finally:
        lib().ptrh_free(out)


def json_dump(text):
    """Parses `text` with the host's JSON reader and returns its compact re-serialisation."""
    raw = text.encode() if isinstance(text, str) else bytes(text)
    out = C.c_void_p()
    if lib().ptrh_json_dump(raw, len(raw), C.byref(out)) != 0:
        raise HostError(lib().ptrh_last_error().decode())
    try:
        return C.string_at(out).decode()
    finally:
        lib().ptrh_free(out)
