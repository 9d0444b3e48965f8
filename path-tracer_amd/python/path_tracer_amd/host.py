"""ctypes binding of host/libptr_host.so: the scene-preparation code (BVH builder + GPU flattener, camera)."""
import ctypes as C

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
        L.ptrh_free.argtypes = [C.c_void_p]
        L.ptrh_camera.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p,
                                  C.c_int32, C.c_void_p]
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


def bvh_flatten(triangles, ranks=None, threads=0):
    """Returns the GPU layout blob (uint8 array): 64-byte header | 64-byte nodes | 64-byte triangles."""
    tri = np.ascontiguousarray(triangles, dtype=np.float32).reshape(-1, 9)
    n = tri.shape[0]
    r = None if ranks is None else np.ascontiguousarray(ranks, dtype=np.int32)
    ptr = C.c_void_p()
    size = C.c_uint64()
    rc = lib().ptrh_bvh_flatten(tri.ctypes.data, n, None if r is None else r.ctypes.data, threads, C.byref(ptr), C.byref(size))
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
