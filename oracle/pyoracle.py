"""TEST INFRASTRUCTURE ONLY: ctypes binding of oracle/liboracle.so (the CPU restatement of the
reference shader path).  Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")


class Texture(C.Structure):
    _fields_ = [("rgba", C.c_void_p), ("w", C.c_uint32), ("h", C.c_uint32)]


class Image(C.Structure):
    _fields_ = [("rgba", C.c_void_p), ("w", C.c_uint32), ("h", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("nodes", C.c_uint64), ("triangles", C.c_uint64), ("shaded", C.c_uint64),
                ("nan_hits", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class Frame(C.Structure):
    _fields_ = [("scene_meta", C.c_void_p), ("bvh_nodes", C.c_void_p), ("indices", C.c_void_p),
                ("positions", C.c_void_p), ("normals", C.c_void_p), ("uvs", C.c_void_p), ("materials", C.c_void_p),
                ("textures", C.POINTER(Texture)), ("texture_count", C.c_uint32), ("accumulation", Image),
                ("output", Image), ("aov_prim", C.c_void_p), ("aov_t", C.c_void_p), ("flat_bvh", C.c_void_p),
                ("use_flat", C.c_int32), ("index_words", C.c_uint32), ("seed_sample_index", C.c_uint32),
                ("flat_instances", C.c_void_p), ("spheres", C.c_void_p), ("sphere_count", C.c_uint32)]


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, "oracle.c")):
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        fp = C.c_void_p
        L.orc_render.argtypes = [C.POINTER(Frame), C.c_uint32, C.c_uint32, C.c_int, C.POINTER(Counters)]
        L.orc_tonemap.argtypes = [C.POINTER(Frame), C.c_uint32, C.c_uint32]
        L.orc_pcg_hash.restype = C.c_uint32
        L.orc_pcg_hash.argtypes = [C.POINTER(C.c_uint32)]
        L.orc_rand.restype = C.c_float
        L.orc_rand.argtypes = [C.POINTER(C.c_uint32)]
        for name in ("orc_sin", "orc_cos"):
            getattr(L, name).restype = C.c_float
            getattr(L, name).argtypes = [C.c_float]
        L.orc_pow.restype = C.c_float
        L.orc_pow.argtypes = [C.c_float, C.c_float]
        L.orc_srgb_to_linear.argtypes = [fp, fp]
        L.orc_linear_to_srgb.argtypes = [fp, fp]
        L.orc_hit_aabb.argtypes = [fp, fp, fp, fp, C.c_float, C.c_float]
        L.orc_hit_triangle_pos.argtypes = [fp, fp, fp, fp, fp, C.c_float, C.c_float, fp, fp, fp]
        L.orc_hit_sphere.argtypes = [fp, C.c_float, fp, fp, C.c_float, C.c_float, fp]
        L.orc_rotation_to_z_axis.argtypes = [fp, fp]
        L.orc_rotate_point.argtypes = [fp, fp, fp]
        L.orc_sample_hemisphere.argtypes = [C.c_float, C.c_float, fp]
        L.orc_disk_vec.argtypes = [C.c_float, C.c_float, fp]
        L.orc_sample_ggx_vndf.argtypes = [fp, C.c_float, C.c_float, C.c_uint32, fp]
        L.orc_brdf_probability.restype = C.c_float
        L.orc_brdf_probability.argtypes = [fp, C.c_float, fp, fp]
        L.orc_sample_texture.argtypes = [C.POINTER(Texture), C.c_float, C.c_float, fp]
        L.orc_camera_ray.argtypes = [fp, C.c_int32, C.c_int32, C.c_uint32, fp, fp]
        L.orc_closest_hit.argtypes = [C.POINTER(Frame), fp, fp, C.c_float, C.c_float, fp, fp, fp, C.POINTER(Counters)]
        L.orc_closest_hit_flat.argtypes = [fp, fp, fp, C.c_float, C.c_float, fp, fp, fp, C.POINTER(Counters)]
        L.orc_closest_hit_two_level.argtypes = [fp, fp, fp, fp, C.c_float, C.c_float, fp, fp, fp, fp, C.POINTER(Counters)]
        L.orc_hit_aabbs.restype = C.c_uint32
        L.orc_hit_aabbs.argtypes = [C.POINTER(Frame), fp, fp, C.c_float, C.c_float]
        _lib = L
    return _lib


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class OracleScene:
    """Holds numpy copies of the scene arrays (so the C side's pointers stay alive) and renders frames.

    `arrays` is a dict with: meta (160-byte uint8 array), nodes (N x 8 uint32 view of packed nodes),
    indices (uint32), positions/normals/uvs (float32), materials (M x 12 uint32/float32 view, 48 B each),
    textures (list of HxWx4 uint8 arrays; index 0 = None), optional flat (uint8 blob)."""

    def __init__(self, arrays):
        self.meta = np.ascontiguousarray(arrays["meta"], dtype=np.uint8).copy()
        assert self.meta.nbytes == 160
        self.nodes = np.ascontiguousarray(arrays["nodes"])
        self.indices = np.ascontiguousarray(arrays["indices"], dtype=np.uint32)
        self.index_words = int(arrays.get("index_words", 3))
        self.seed_sample_index = 0   # see orc_frame.seed_sample_index
        self.positions = f32(arrays["positions"])
        self.normals = f32(arrays["normals"])
        self.uvs = f32(arrays["uvs"])
        self.materials = np.ascontiguousarray(arrays["materials"])
        assert self.materials.nbytes % 48 == 0
        self.textures = [None if t is None else np.ascontiguousarray(t, dtype=np.uint8) for t in arrays["textures"]]
        self.flat = None if arrays.get("flat") is None else np.ascontiguousarray(arrays["flat"], dtype=np.uint8)
        self.instances = None if arrays.get("flat_instances") is None else np.ascontiguousarray(arrays["flat_instances"], dtype=np.uint8)
        # sphere leaves: the reference's 80-byte `sphere` records (compute.comp:31-36) the flattened layout's sphere leaves name
        self.spheres = None if arrays.get("sphere_records") is None else np.ascontiguousarray(arrays["sphere_records"])
        assert self.spheres is None or self.spheres.nbytes % 80 == 0
        self._tex = (Texture * max(1, len(self.textures)))()
        for i, t in enumerate(self.textures):
            if t is not None:
                self._tex[i] = Texture(t.ctypes.data, t.shape[1], t.shape[0])

    def meta_u32(self):
        return self.meta.view(np.uint32)

    def set_meta(self, **kw):
        names = dict(max_bounce=32, min_bounce=33, width=34, height=35, sample_index=36, enable_dof=37, debug_bvh=38,
                     downscale_factor=39)
        u = self.meta.view(np.uint32)
        for k, v in kw.items():
            u[names[k]] = np.uint32(v)

    def frame(self, acc, out, aov=None, use_flat=False):
        f = Frame()
        f.scene_meta = self.meta.ctypes.data
        f.bvh_nodes = self.nodes.ctypes.data
        f.indices = self.indices.ctypes.data
        f.index_words = self.index_words
        f.seed_sample_index = self.seed_sample_index
        f.positions = self.positions.ctypes.data
        f.normals = self.normals.ctypes.data
        f.uvs = self.uvs.ctypes.data
        f.materials = self.materials.ctypes.data
        f.textures = self._tex
        f.texture_count = len(self.textures)
        f.accumulation = Image(acc.ctypes.data, acc.shape[1], acc.shape[0])
        f.output = Image(out.ctypes.data, out.shape[1], out.shape[0])
        if aov is not None:
            f.aov_prim = aov[0].ctypes.data
            f.aov_t = aov[1].ctypes.data
        if use_flat:
            assert self.flat is not None
            f.flat_bvh = self.flat.ctypes.data
            f.use_flat = 1
            if self.instances is not None:
                f.flat_instances = self.instances.ctypes.data
            if self.spheres is not None:
                f.spheres = self.spheres.ctypes.data
                f.sphere_count = self.spheres.nbytes // 80
        return f

    def render(self, acc, out, rows=None, threads=None, aov=None, use_flat=False):
        """One compute.comp dispatch.  acc/out: (H, W, 4) float32 arrays.  Returns the counters."""
        u = self.meta_u32()
        h = int(u[35])
        y0, y1 = rows if rows is not None else (0, h)
        c = Counters()
        f = self.frame(acc, out, aov, use_flat)
        lib().orc_render(C.byref(f), y0, y1, threads or (os.cpu_count() or 1), C.byref(c))
        return c.as_dict()

    def tonemap(self, acc, out, groups_x, groups_y):
        f = self.frame(acc, out)
        lib().orc_tonemap(C.byref(f), groups_x, groups_y)
