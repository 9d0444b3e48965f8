"""ctypes binding of csrc/libptr_hip.so (the C ABI of include/ptr_hip.h) plus thin Python mirrors of
the reference's renderer-side classes for the compute path:

  Context            the ptr_ctx: one GPU, one stream
  Buffer / Texture / Sampler   shapes of include/vk-renderer.hpp:23-38,160-228 in the reference
  ComputeRenderpass  same methods as include/compute-renderpass.hpp:9-41 (set_pipeline, set_dispatch_size,
                     set_ouput_texture [sic], set_constant, execute)
  GpuScene           uploads flat scene arrays the way src/scene.cpp:183-204 does and wires the seven
                     buffer constants the way src/main.cpp:263-270 does

There is no CPU fallback: without the built HIP library every constructor here raises."""
import ctypes as C
import os

import numpy as np

from . import paths, scenes

FORMAT_RGBA8_UNORM = 37
FORMAT_RGBA32_SFLOAT = 109
FILTER_LINEAR = 1
ADDRESS_REPEAT = 0


class HipError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(paths.HIP_LIB):
        raise HipError(f"{paths.HIP_LIB} is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950); there is no CPU fallback for the render path")
    L = C.CDLL(paths.HIP_LIB)
    vp, u32, u64, sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_size_t
    P = C.POINTER
    sigs = {
        "ptr_ctx_create": [C.c_int, P(vp)],
        "ptr_ctx_destroy": [vp],
        "ptr_ctx_set_stream": [vp, vp],
        "ptr_sync": [vp],
        "ptr_flush": [vp],
        "ptr_buffer_create": [vp, sz, P(u32)],
        "ptr_buffer_destroy": [vp, u32],
        "ptr_buffer_write": [vp, u32, vp, sz, sz],
        "ptr_buffer_read": [vp, u32, vp, sz, sz],
        "ptr_buffer_address": [vp, u32, P(u64)],
        "ptr_sampler_create": [vp, C.c_int, C.c_int, P(u32)],
        "ptr_image_create": [vp, u32, u32, C.c_int, u32, P(u32)],
        "ptr_image_wrap": [vp, vp, u32, u32, C.c_int, P(u32)],
        "ptr_image_destroy": [vp, u32],
        "ptr_image_upload": [vp, u32, vp],
        "ptr_image_read": [vp, u32, vp],
        "ptr_image_read_unorm8": [vp, u32, vp],
        "ptr_image_index": [vp, u32, P(u32)],
        "ptr_sampler_index": [vp, u32, P(u32)],
        "ptr_image_address": [vp, u32, P(u64)],
        "ptr_bvh_bind_flat": [vp, u32, u32],
        "ptr_bvh_bind_instances": [vp, u32, u32],
        "ptr_bvh_bind_spheres": [vp, u32, u32, u32],
        "ptr_dispatch": [vp, C.c_char_p, vp, u32, u32, u32],
        "ptr_dispatch_static": [vp, C.c_char_p, vp, u32, u32, u32],
        "ptr_set_partition": [vp, u32, u32, u32],
        "ptr_image_pack_stripes": [vp, u32, u32, u32, u32, u32],
        "ptr_image_unpack_stripes": [vp, u32, u32, u32, u32],
        "ptr_set_sample_partition": [vp, u32, u32],
        "ptr_set_option": [vp, C.c_char_p, C.c_int],
        "ptr_set_capture": [vp, u32],
        "ptr_stats_enable": [vp, C.c_int],
        "ptr_stats_read": [vp, P(u64), C.c_int],
        "ptr_profile_enable": [vp, C.c_int],
        "ptr_profile_read": [vp, C.c_char_p, P(u64), P(C.c_double), C.c_int],
        "ptr_profile_names": [vp, C.c_char_p, sz],
        "ptr_device_function": [vp, C.c_char_p, vp, u32, vp, u32, u32],
        "ptr_bvh_build_flat": [vp, u32, u32, u32, P(u32)],
        "ptr_bvh_flat_set_reference_ranks": [vp, u32, u32, u32],
    }
    for name, args in sigs.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = C.c_int
    L.ptr_last_error.argtypes = [vp]
    L.ptr_last_error.restype = C.c_char_p
    _lib = L
    return L


ABI_SYMBOLS = ["ptr_ctx_create", "ptr_ctx_destroy", "ptr_ctx_set_stream", "ptr_last_error", "ptr_sync", "ptr_flush", "ptr_buffer_create",
               "ptr_buffer_destroy", "ptr_buffer_write", "ptr_buffer_read", "ptr_buffer_address", "ptr_sampler_create",
               "ptr_image_create", "ptr_image_wrap", "ptr_image_destroy", "ptr_image_upload", "ptr_image_read", "ptr_image_read_unorm8",
               "ptr_image_index", "ptr_sampler_index", "ptr_image_address", "ptr_bvh_bind_flat", "ptr_bvh_bind_instances", "ptr_bvh_bind_spheres", "ptr_dispatch", "ptr_dispatch_static",
               "ptr_set_partition", "ptr_image_pack_stripes", "ptr_image_unpack_stripes", "ptr_set_sample_partition", "ptr_set_option", "ptr_set_capture", "ptr_stats_enable", "ptr_stats_read", "ptr_profile_enable",
               "ptr_profile_read", "ptr_profile_names", "ptr_device_function", "ptr_bvh_build_flat", "ptr_bvh_flat_set_reference_ranks"]


class Context:
    def __init__(self, device=0, stream=None):
        self._L = lib()
        h = C.c_void_p()
        rc = self._L.ptr_ctx_create(device, C.byref(h))
        if rc != 0:
            raise HipError(f"ptr_ctx_create({device}) = {rc}: {self._L.ptr_last_error(None).decode()}")
        self.handle = h
        self.device = device
        if stream is not None:
            self.set_stream(stream)

    def _check(self, rc):
        if rc != 0:
            raise HipError(f"[{rc}] {self._L.ptr_last_error(self.handle).decode()}")

    def close(self):
        if self.handle:
            self._L.ptr_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        self._check(self._L.ptr_ctx_set_stream(self.handle, C.c_void_p(stream_ptr)))

    def sync(self):
        self._check(self._L.ptr_sync(self.handle))

    def flush(self):
        """Enqueues the dispatches held back under "dispatch.coalesce" without waiting for them."""
        self._check(self._L.ptr_flush(self.handle))

    # ---- factories named after the reference's vkrenderer statics ----
    def create_buffer(self, size):
        return Buffer(self, size)

    def create_sampler(self, filter=FILTER_LINEAR, address_mode=ADDRESS_REPEAT):
        return Sampler(self, filter, address_mode)

    def create_2d_texture(self, width, height, format=FORMAT_RGBA8_UNORM, sampler=None):
        return Texture(self, width, height, format, sampler)

    def wrap_2d_texture(self, device_ptr, width, height, format=FORMAT_RGBA32_SFLOAT):
        return Texture(self, width, height, format, None, device_ptr=device_ptr)

    def create_compute_renderpass(self):
        return ComputeRenderpass(self)

    # ---- instrumentation ----
    def set_partition(self, rank, world, stripe_rows=8):
        self._check(self._L.ptr_set_partition(self.handle, rank, world, stripe_rows))

    def pack_stripes(self, image, rank, world, stripe_rows, packed_buffer):
        self._check(self._L.ptr_image_pack_stripes(self.handle, image.handle, rank, world, stripe_rows, packed_buffer.handle))

    def unpack_stripes(self, image, world, stripe_rows, gathered_buffer):
        self._check(self._L.ptr_image_unpack_stripes(self.handle, image.handle, world, stripe_rows, gathered_buffer.handle))

    def set_sample_partition(self, rank, world):
        self._check(self._L.ptr_set_sample_partition(self.handle, rank, world))

    def set_option(self, key, value):
        self._check(self._L.ptr_set_option(self.handle, key.encode(), int(value)))

    def set_capture(self, buffer):
        self._check(self._L.ptr_set_capture(self.handle, buffer.handle if buffer is not None else 0))

    def stats_enable(self, on):
        self._check(self._L.ptr_stats_enable(self.handle, 1 if on else 0))

    def stats_read(self, reset=True):
        out = (C.c_uint64 * 8)()
        self._check(self._L.ptr_stats_read(self.handle, out, 1 if reset else 0))
        return dict(rays=int(out[0]), nodes=int(out[1]), triangles=int(out[2]), shaded=int(out[3]), frames=int(out[4]),
                    wave_inner_steps=int(out[5]), wave_leaf_steps=int(out[6]), idle_lane_steps=int(out[7]))

    def profile_enable(self, on):
        self._check(self._L.ptr_profile_enable(self.handle, 1 if on else 0))

    def profile_read(self, reset=True):
        buf = C.create_string_buffer(4096)
        self._check(self._L.ptr_profile_names(self.handle, buf, 4096))
        out = {}
        for name in [n for n in buf.value.decode().split("\n") if n]:
            n = C.c_uint64()
            ms = C.c_double()
            self._check(self._L.ptr_profile_read(self.handle, name.encode(), C.byref(n), C.byref(ms), 1 if reset else 0))
            out[name] = dict(launches=int(n.value), total_ms=float(ms.value))
        return out

    def device_function(self, name, records, out_words):
        """Evaluates one device function on `records` ((count, in_words) array of 32-bit words: float32 or uint32,
        or a structured mix viewed as uint32); returns a (count, out_words) uint32 array (view as float32 as needed)."""
        a = np.ascontiguousarray(records)
        assert a.ndim == 2 and a.dtype.itemsize == 4
        out = np.zeros((a.shape[0], out_words), np.uint32)
        self._check(self._L.ptr_device_function(self.handle, name.encode(), a.ctypes.data, a.shape[1], out.ctypes.data, out_words, a.shape[0]))
        return out

    def build_flat(self, indices_buffer, positions_buffer, triangle_count):
        """GPU build of the flattened BVH from the scene's device buffers; returns the new Buffer."""
        h = C.c_uint32()
        self._check(self._L.ptr_bvh_build_flat(self.handle, indices_buffer.handle, positions_buffer.handle, triangle_count, C.byref(h)))
        return Buffer.adopt(self, h.value, 64 + 128 * triangle_count)

    def flat_set_reference_ranks(self, flat_buffer, node_buffer, triangle_count):
        """Tie-break ranks of the reference's visiting order (the leaf order of the threaded nodes) for a flattened layout."""
        self._check(self._L.ptr_bvh_flat_set_reference_ranks(self.handle, flat_buffer.handle, node_buffer.handle, triangle_count))

    def bind_flat(self, node_buffer, flat_buffer):
        self._check(self._L.ptr_bvh_bind_flat(self.handle, node_buffer.handle, flat_buffer.handle if flat_buffer is not None else 0))

    def bind_instances(self, flat_buffer, instance_buffer):
        """Top level over placed copies of the layout in flat_buffer (host.bvh_flatten_instances); None removes it."""
        self._check(self._L.ptr_bvh_bind_instances(self.handle, flat_buffer.handle, instance_buffer.handle if instance_buffer is not None else 0))

    def bind_spheres(self, flat_buffer, sphere_buffer, count=0):
        """The 80-byte sphere records (compute.comp:31-36) named by the sphere leaves of the layout in flat_buffer
        (host.bvh_flatten(..., spheres=...)); None removes them."""
        self._check(self._L.ptr_bvh_bind_spheres(self.handle, flat_buffer.handle, sphere_buffer.handle if sphere_buffer is not None else 0, count))

    def dispatch(self, kernel, constants, gx, gy, gz=1):
        k = np.ascontiguousarray(constants, dtype=np.uint8)
        assert k.nbytes == 64
        self._check(self._L.ptr_dispatch(self.handle, kernel.encode(), k.ctypes.data, gx, gy, gz))

    def dispatch_static(self, kernel, constants, gx, gy, gz=1):
        """ptr_dispatch_static: the same frame in the library's fixed per-GPU scratch array, nothing allocated."""
        k = np.ascontiguousarray(constants, dtype=np.uint8)
        assert k.nbytes == 64
        self._check(self._L.ptr_dispatch_static(self.handle, kernel.encode(), k.ctypes.data, gx, gy, gz))


class Buffer:
    def __init__(self, ctx, size):
        self.ctx = ctx
        h = C.c_uint32()
        ctx._check(ctx._L.ptr_buffer_create(ctx.handle, size, C.byref(h)))
        self.handle = h.value
        self.buffer_size = size

    @classmethod
    def adopt(cls, ctx, handle, size=None):
        """A buffer the library created itself (ptr_bvh_build_flat)."""
        b = cls.__new__(cls)
        b.ctx, b.handle, b.buffer_size = ctx, handle, size
        return b

    def write(self, data, alloc_offset=0, data_size=None):
        a = np.ascontiguousarray(data)
        n = a.nbytes if data_size is None else data_size
        self.ctx._check(self.ctx._L.ptr_buffer_write(self.ctx.handle, self.handle, a.ctypes.data, alloc_offset, n))

    def read(self, dtype=np.uint8, offset=0, size=None):
        n = self.buffer_size - offset if size is None else size
        out = np.empty(n, dtype=np.uint8)
        self.ctx._check(self.ctx._L.ptr_buffer_read(self.ctx.handle, self.handle, out.ctypes.data, offset, n))
        return out.view(dtype)

    @property
    def device_address(self):
        a = C.c_uint64()
        self.ctx._check(self.ctx._L.ptr_buffer_address(self.ctx.handle, self.handle, C.byref(a)))
        return a.value

    def destroy(self):
        self.ctx._check(self.ctx._L.ptr_buffer_destroy(self.ctx.handle, self.handle))
        self.handle = 0


class Sampler:
    def __init__(self, ctx, filter, address_mode):
        h = C.c_uint32()
        ctx._check(ctx._L.ptr_sampler_create(ctx.handle, filter, address_mode, C.byref(h)))
        self.handle = h.value
        self.bindless_index = h.value


class Texture:
    def __init__(self, ctx, width, height, format, sampler, device_ptr=None):
        self.ctx = ctx
        self.width, self.height, self.depth, self.format, self.sampler = width, height, 1, format, sampler
        h = C.c_uint32()
        if device_ptr is None:
            ctx._check(ctx._L.ptr_image_create(ctx.handle, width, height, format, sampler.handle if sampler else 0, C.byref(h)))
        else:
            ctx._check(ctx._L.ptr_image_wrap(ctx.handle, C.c_void_p(device_ptr), width, height, format, C.byref(h)))
        self.handle = h.value
        self.bindless_index = h.value

    def update(self, data):
        a = np.ascontiguousarray(data)
        expect = self.width * self.height * (4 if self.format == FORMAT_RGBA8_UNORM else 16)
        if a.nbytes != expect:
            raise HipError(f"Texture.update: {a.nbytes} bytes for a {self.width}x{self.height} image of {expect} bytes")
        self.ctx._check(self.ctx._L.ptr_image_upload(self.ctx.handle, self.handle, a.ctypes.data))

    def read(self):
        if self.format == FORMAT_RGBA32_SFLOAT:
            out = np.empty((self.height, self.width, 4), dtype=np.float32)
        else:
            out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        self.ctx._check(self.ctx._L.ptr_image_read(self.ctx.handle, self.handle, out.ctypes.data))
        return out

    def read_unorm8(self):
        """The image as a present shows it: (height, width, 4) uint8, converted on the device (ptr_image_read_unorm8)."""
        out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        self.ctx._check(self.ctx._L.ptr_image_read_unorm8(self.ctx.handle, self.handle, out.ctypes.data))
        return out

    def destroy(self):
        self.ctx._check(self.ctx._L.ptr_image_destroy(self.ctx.handle, self.handle))
        self.handle = 0


class ComputeRenderpass:
    """Mirror of the reference's ComputeRenderpass (include/compute-renderpass.hpp:9-41): a 64-byte
    constant block, a dispatch size and an output texture; execute() pushes the block and dispatches."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.constants = np.zeros(64, dtype=np.uint8)
        self.group_count = (0, 0, 0)
        self.pipeline = None
        self.output_texture = None

    def set_pipeline(self, shader_name):
        if shader_name not in ("compute", "compute.megakernel", "compute.wavefront", "tonemapping", "linearize", "resolve"):
            raise HipError(f"set_pipeline: unknown shader '{shader_name}'")
        self.pipeline = shader_name

    def set_dispatch_size(self, count_x, count_y, count_z):
        self.group_count = (count_x, count_y, count_z)

    def set_ouput_texture(self, out_texture):  # spelling as in the reference
        self.output_texture = out_texture
        self.constants[56:60] = np.frombuffer(np.uint32(out_texture.bindless_index).tobytes(), dtype=np.uint8)

    def set_constant(self, offset, value):
        if isinstance(value, Buffer):
            self.constants[offset:offset + 8] = np.frombuffer(np.uint64(value.device_address).tobytes(), dtype=np.uint8)
        elif isinstance(value, Texture):
            self.constants[offset:offset + 4] = np.frombuffer(np.uint32(value.bindless_index).tobytes(), dtype=np.uint8)
        else:
            self.constants[offset:offset + 8] = np.frombuffer(np.uint64(value).tobytes(), dtype=np.uint8)

    def execute(self):
        if self.pipeline is None:
            raise HipError("execute: no pipeline set")
        self.ctx.dispatch(self.pipeline, self.constants, *self.group_count)


class GpuScene:
    """Device-resident scene: the seven buffers of src/scene.cpp:183-204 + the RGBA8 textures, from flat
    arrays (see scenes.py).  `nodes` is the threaded node array, `flat` (optional) the flattened layout."""

    def __init__(self, ctx, arrays, nodes, flat=None, meta=None):
        self.ctx = ctx
        self.sampler = ctx.create_sampler(FILTER_LINEAR, ADDRESS_REPEAT)
        self.textures = [None]
        tex_index = {0: 0}
        for i, t in enumerate(arrays["textures"]):
            if t is None:
                continue
            tex = ctx.create_2d_texture(t.shape[1], t.shape[0], FORMAT_RGBA8_UNORM, self.sampler)
            tex.update(t)
            self.textures.append(tex)
            tex_index[i] = tex.bindless_index
        mats = np.array(arrays["materials"], copy=True)
        # materials carry scene-local texture numbers; translate them to this context's bindless indices
        mats["albedo_tex"] = [tex_index[int(x)] for x in mats["albedo_tex"]]
        mats["mr_tex"] = [tex_index[int(x)] for x in mats["mr_tex"]]
        mats["albedo_sampler"] = np.where(mats["albedo_tex"] > 0, self.sampler.bindless_index, 0)
        mats["mr_sampler"] = np.where(mats["mr_tex"] > 0, self.sampler.bindless_index, 0)
        self.materials_host = mats
        self.texture_table = [None] + [None] * max(tex_index.values())
        for i, t in enumerate(arrays["textures"]):
            if t is not None:
                self.texture_table[tex_index[i]] = t

        def upload(a):
            a = np.ascontiguousarray(a)
            b = ctx.create_buffer(max(a.nbytes, 16))
            b.write(a)
            return b
        self.scene_buffer = ctx.create_buffer(160 * 2)
        self.bvh_buffer = upload(nodes)
        self.index_words = int(arrays.get("index_words", 3))
        self.indices_buffer = upload(np.asarray(arrays["indices"], dtype=np.uint32))
        self.positions_buffer = upload(np.asarray(arrays["positions"], dtype=np.float32))
        self.normals_buffer = upload(np.asarray(arrays["normals"], dtype=np.float32))
        self.uvs_buffer = upload(np.asarray(arrays["uvs"], dtype=np.float32))
        self.materials_buffer = upload(mats)
        self.flat_buffer = None
        if flat is not None:
            self.flat_buffer = upload(flat)
            ctx.bind_flat(self.bvh_buffer, self.flat_buffer)
        self.instance_buffer = None
        if flat is not None and arrays.get("flat_instances") is not None:
            self.instance_buffer = upload(arrays["flat_instances"])
            ctx.bind_instances(self.flat_buffer, self.instance_buffer)
        self.sphere_buffer = None
        self.spheres_host = None
        if flat is not None and arrays.get("sphere_records") is not None:
            records = np.array(arrays["sphere_records"], dtype=scenes.SPHERE_DTYPE, copy=True)
            m = records["material"]   # the embedded materials name textures the same way the material buffer does
            m["albedo_tex"] = [tex_index[int(x)] for x in m["albedo_tex"]]
            m["mr_tex"] = [tex_index[int(x)] for x in m["mr_tex"]]
            m["albedo_sampler"] = np.where(m["albedo_tex"] > 0, self.sampler.bindless_index, 0)
            m["mr_sampler"] = np.where(m["mr_tex"] > 0, self.sampler.bindless_index, 0)
            self.spheres_host = records
            self.sphere_buffer = upload(records)
            ctx.bind_spheres(self.flat_buffer, self.sphere_buffer, len(records))
        self.meta = None
        if meta is not None:
            self.set_meta(meta)

    def set_meta(self, meta):
        self.meta = np.ascontiguousarray(meta, dtype=np.uint8).copy()
        self.scene_buffer.write(self.meta, 0, 160)

    def update_meta(self, **kw):
        names = dict(max_bounce=32, min_bounce=33, width=34, height=35, sample_index=36, enable_dof=37, debug_bvh=38,
                     downscale_factor=39)
        u = self.meta.view(np.uint32)
        for k, v in kw.items():
            u[names[k]] = np.uint32(v)
        self.scene_buffer.write(self.meta, 0, 160)

    def wire(self, renderpass):
        """src/main.cpp:263-269"""
        self.ctx.set_option("scene.index_words", self.index_words)   # a context-wide setting: the layout of this scene's index buffer
        renderpass.set_constant(0, self.scene_buffer)
        renderpass.set_constant(8, self.bvh_buffer)
        renderpass.set_constant(16, self.indices_buffer)
        renderpass.set_constant(24, self.positions_buffer)
        renderpass.set_constant(32, self.normals_buffer)
        renderpass.set_constant(40, self.uvs_buffer)
        renderpass.set_constant(48, self.materials_buffer)


class FrameLoop:
    """The reference's per-frame protocol (src/main.cpp:406-416) without a window: set the output texture
    and the accumulation constant, execute, advance sample_index, swap the two images."""

    def __init__(self, ctx, scene, width, height, pipeline="compute", textures=None, samples_per_dispatch=1):
        self.ctx, self.scene, self.width, self.height = ctx, scene, width, height
        # > 1: one dispatch renders that many consecutive samples ("dispatch.samples", wavefront pipeline only); the images
        # swap once per dispatch and the result equals that many single-sample frames
        self.samples_per_dispatch = samples_per_dispatch
        if textures is None:
            self.accumulation = ctx.create_2d_texture(width, height, FORMAT_RGBA32_SFLOAT)
            self.output = ctx.create_2d_texture(width, height, FORMAT_RGBA32_SFLOAT)
        else:
            self.accumulation, self.output = textures
        self.renderpass = ctx.create_compute_renderpass()
        self.renderpass.set_pipeline(pipeline)
        scene.wire(self.renderpass)
        self.renderpass.set_dispatch_size(width // 8 + 1, height // 8 + 1, 1)  # src/main.cpp:270
        self.sample_index = 1

    def reset(self):
        self.sample_index = 1

    def frame(self, samples=1):
        """One dispatch of `samples` consecutive samples (1 = the reference's protocol)."""
        self.ctx.set_option("dispatch.samples", samples)
        self.scene.update_meta(sample_index=self.sample_index)
        self.renderpass.set_ouput_texture(self.output)
        self.renderpass.set_constant(60, self.accumulation)
        try:
            self.renderpass.execute()
        finally:
            if samples != 1:
                self.ctx.set_option("dispatch.samples", 1)
        self.sample_index += samples
        self.output, self.accumulation = self.accumulation, self.output

    def frames(self, count, schedule=None):
        """`count` samples, `samples_per_dispatch` at a time (the last dispatch takes what is left); or, with `schedule`,
        as dispatches of exactly those sizes (they must add up to `count`).  The image is the same either way."""
        if schedule is not None:
            if sum(schedule) != count or min(schedule) < 1:
                raise ValueError(f"dispatch sizes {schedule} do not add up to {count} samples")
            for n in schedule:
                self.frame(n)
            return
        while count > 0:
            n = min(count, self.samples_per_dispatch)
            self.frame(n)
            count -= n

    def latest(self):
        """The image the last frame wrote (after the swap it is `accumulation`)."""
        return self.accumulation
