/* C ABI of the MI355X render hot path (libptr_hip.so).
 *
 * This is the boundary the reference's compute renderpass would bind instead of Vulkan: every entry
 * point replaces one piece of the reference's renderer interface for the `compute.comp` /
 * `tonemapping.comp` dispatch path (citations are files of the reference repository).  Plain pointers,
 * sizes and integer handles only; every call returns 0 on success or a negative PTR_E_* code
 * (ptr_last_error() gives the text).  A context is bound to one GPU and is NOT thread-safe, like the
 * reference's single-threaded renderer (include/vk-api.hpp:212-214 keeps plain vectors, no locks).
 */
#ifndef PTR_HIP_H_
#define PTR_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ptr_ctx ptr_ctx;
typedef uint32_t ptr_handle; /* 0 is never a valid handle */

enum {
    PTR_OK = 0,
    PTR_E_INVALID = -1,     /* bad argument / unknown handle / unknown kernel name */
    PTR_E_HIP = -2,         /* a HIP runtime call failed */
    PTR_E_RANGE = -3,       /* offset/size outside the resource */
    PTR_E_UNSUPPORTED = -4, /* format / sampler mode the path does not implement */
    PTR_E_STATE = -5        /* e.g. a dispatch whose constant block names no registered resource */
};

/* image formats (the two the path uses: VK_FORMAT_R8G8B8A8_UNORM textures, src/gltf.cpp:126,151;
 * VK_FORMAT_R32G32B32A32_SFLOAT accumulation/output images, src/main.cpp:261-262) */
enum { PTR_FORMAT_RGBA8_UNORM = 37, PTR_FORMAT_RGBA32_SFLOAT = 109 }; /* numeric values = VkFormat */
/* sampler state (VkFilter / VkSamplerAddressMode values); the path implements LINEAR + REPEAT, the only
 * sampler the reference creates (src/gltf.cpp:118, src/vk-api.cpp:406-424) */
enum { PTR_FILTER_NEAREST = 0, PTR_FILTER_LINEAR = 1 };
enum { PTR_ADDRESS_REPEAT = 0, PTR_ADDRESS_CLAMP_TO_EDGE = 2 };

/* ---- context: replaces the vkcontext/vkapi singletons (src/vk-renderer.cpp:21-23, src/vk-context.cpp:147-199) ---- */
int ptr_ctx_create(int device, ptr_ctx** out);
int ptr_ctx_destroy(ptr_ctx* ctx);
/* Launch on a caller-owned hipStream_t (NULL = the context's own stream). */
int ptr_ctx_set_stream(ptr_ctx* ctx, void* hip_stream);
/* Text of the last error on this context (or of context creation when ctx is NULL). */
const char* ptr_last_error(const ptr_ctx* ctx);
/* Blocks until everything enqueued on the context's stream has finished (vkrenderer::begin_frame's
 * fence wait, src/vk-renderer.cpp:214-221). */
int ptr_sync(ptr_ctx* ctx);
/* Enqueues the "compute" dispatches the context is holding back ("dispatch.coalesce" below) without waiting for them: for
 * callers that go on to use the images on the same stream by means of their own (a wrapped image, a device pointer).  Every
 * other entry point that reads or changes what a dispatch sees does the same on its own.  No counterpart in the reference,
 * whose dispatches are recorded into the frame's command buffer as they come (src/compute-renderpass.cpp:51-56). */
int ptr_flush(ptr_ctx* ctx);

/* ---- buffers: vkrenderer::create_buffer + Buffer::write (src/vk-renderer.cpp:140-142,248-261) and the
 *      device address the renderpass pushes (src/compute-renderpass.cpp:46-49) ---- */
int ptr_buffer_create(ptr_ctx* ctx, size_t size, ptr_handle* out);
int ptr_buffer_destroy(ptr_ctx* ctx, ptr_handle buffer);
int ptr_buffer_write(ptr_ctx* ctx, ptr_handle buffer, const void* data, size_t offset, size_t size);
int ptr_buffer_read(ptr_ctx* ctx, ptr_handle buffer, void* dst, size_t offset, size_t size);
int ptr_buffer_address(ptr_ctx* ctx, ptr_handle buffer, uint64_t* device_address);

/* ---- samplers and images: vkrenderer::create_sampler / create_2d_texture / Texture::update
 *      (include/vk-renderer.hpp:64-66,166-228) and the bindless indices of vkapi (include/vk-api.hpp:72-86).
 *      Index 0 is "no texture": sampling it yields opaque white. ---- */
int ptr_sampler_create(ptr_ctx* ctx, int filter, int address_mode, ptr_handle* out);
int ptr_image_create(ptr_ctx* ctx, uint32_t width, uint32_t height, int format, ptr_handle sampler, ptr_handle* out);
/* Same, over device memory the caller owns (width*height*texel bytes, 16-byte aligned); never freed by the library. */
int ptr_image_wrap(ptr_ctx* ctx, void* device_memory, uint32_t width, uint32_t height, int format, ptr_handle* out);
int ptr_image_destroy(ptr_ctx* ctx, ptr_handle image);
int ptr_image_upload(ptr_ctx* ctx, ptr_handle image, const void* texels);  /* whole image, host memory */
int ptr_image_read(ptr_ctx* ctx, ptr_handle image, void* texels);          /* whole image, host memory */
/* The texels a present would show, width * height RGBA8 words (R in the low byte): the reference blits the RGBA32F output image
 * to the swapchain's 8-bit UNORM image every frame (vkapi::blit_full, src/vk-api.cpp:1194-1224, called at
 * src/vk-renderer.cpp:199); the conversion is the API's float -> UNORM rule -- NaN 0, clamp to [0, 1], x 255, round to nearest --
 * and runs on the device, so a caller that shows every frame moves a quarter of the bytes ptr_image_read moves. */
int ptr_image_read_unorm8(ptr_ctx* ctx, ptr_handle image, void* rgba8);
int ptr_image_index(ptr_ctx* ctx, ptr_handle image, uint32_t* bindless_index);
int ptr_sampler_index(ptr_ctx* ctx, ptr_handle sampler, uint32_t* bindless_index);
int ptr_image_address(ptr_ctx* ctx, ptr_handle image, uint64_t* device_address);

/* ---- acceleration layout: associates the flattened, cache-line-aligned BVH blob emitted by
 *      bvh::flatten() (a buffer holding header|nodes|triangles) with the threaded node buffer the
 *      constant block names at offset 8.  A later "compute" dispatch on that node buffer then runs
 *      the wavefront kernels over the flattened layout.  flat = 0 removes the association. ---- */
int ptr_bvh_bind_flat(ptr_ctx* ctx, ptr_handle threaded_nodes, ptr_handle flat);

/* Placed copies of a flattened layout (instancing) -- an extension: the reference's scene is one triangle soup
 * (src/scene.cpp:84-176), so a scene of repeated objects is written out copy by copy there.  `instances` holds the blob
 * bvh::flatten_instances() / ptrh_bvh_flatten_instances() emit for the layout in `flat`: header | 64-byte nodes over the
 * copies' world-space boxes | one 64-byte record per copy (three rows of the world-to-object matrix, the object's root).
 * A later wavefront dispatch over `flat` walks that top level first and enters `flat` once per copy the ray reaches, with
 * the ray taken into the object's space; normals come back through the transposed matrix.  The index / vertex / material
 * buffers of the constant block stay the object's own.  A captured primitive id (ptr_set_capture) reads
 * copy * triangle_count + primitive.  instances = 0 removes the association.  The megakernel path does not instance. */
int ptr_bvh_bind_instances(ptr_ctx* ctx, ptr_handle flat, ptr_handle instances);

/* Sphere leaves.  The reference's first scene was made of spheres: shaders/include/ray.h:105-135 still holds hit_sphere, its call
 * site in hit_node is commented out (ray.h:173-177, `bufs.geometry.spheres[primitive_id]`), and so are the lines that would give
 * a hit its point and outward normal (ray.h:127-132).  This binds `count` records of the shader's `sphere` struct
 * (compute.comp:31-36: vec4 position, the 48-byte material, float radius, 12 bytes of padding = 80 bytes) to a flattened layout
 * whose primitive array holds sphere records beside its triangles (bvh::flatten(..., spheres) / ptrh_bvh_flatten_spheres).  A
 * sphere is tested with hit_sphere against the ray shortened to the closest hit so far; a hit is shaded with the frame those
 * commented lines describe -- the point on the ray, the normal (point - centre) / radius -- texture coordinates (0, 0) and the
 * material embedded in the record; a captured primitive id reads 0x40000000 | sphere index.  spheres = 0 removes the
 * association.  Wavefront pipeline only, and not together with placed copies. */
int ptr_bvh_bind_spheres(ptr_ctx* ctx, ptr_handle flat, ptr_handle spheres, uint32_t count);

/* ---- dispatch: ComputeRenderpass::execute -> vkCmdPushConstants(64 B) + vkCmdDispatch
 *      (src/compute-renderpass.cpp:51-63, src/vk-api.cpp:1090-1092,1171-1177).
 *      kernel: the reference's shader names, "compute" (shaders/compute.comp) or "tonemapping"
 *      (shaders/tonemapping.comp).  "compute" picks the wavefront path when a flattened layout is
 *      bound to the node buffer and the megakernel over the threaded nodes otherwise;
 *      "compute.megakernel" / "compute.wavefront" force one.
 *      constants: the 64-byte push-constant block (compute.comp:109-120): seven device addresses at
 *      0,8,...,48 (scene, bvh, indices, positions, normals, uvs, materials), output image index at 56,
 *      accumulation image index at 60.  groups_*: counts of 8x8x1 workgroups. ---- */
/* Builds the flattened layout on the GPU from the scene's own device buffers: `indices` (in the layout "scene.index_words"
 * names) and `positions`, `triangle_count` triangles.  *flat_out is a new buffer (destroy it with ptr_buffer_destroy) ready for
 * ptr_bvh_bind_flat.  Extends bvh::bvh (src/bvh.cpp:48-86), which builds on one host thread: a Morton-order radix tree collapsed
 * to the same 4-wide quantised format the host flattener emits -- milliseconds per million triangles, a somewhat slower tree
 * to traverse than the host's binned-SAH one.  Equal-t ties are broken by primitive id (the host path can carry the reference's
 * visiting order instead).  Synchronous.  PTR_E_RANGE when the tree would be deeper than the traversal stack. */
int ptr_bvh_build_flat(ptr_ctx* ctx, ptr_handle indices, ptr_handle positions, uint32_t triangle_count, ptr_handle* flat_out);
/* Gives the triangles of a flattened layout (typically one from ptr_bvh_build_flat, whose ties go by primitive id) the tie-break
 * ranks of the reference: on equal t the reference keeps the triangle it visits later (shaders/include/ray.h:153), and it visits
 * leaves in the order of the threaded node array `threaded_nodes` (include/bvh.hpp:17-22, 2 * triangle_count - 1 records).
 * After this call a dispatch over `flat` resolves ties like the megakernel over those nodes does.  Synchronous. */
int ptr_bvh_flat_set_reference_ranks(ptr_ctx* ctx, ptr_handle flat, ptr_handle threaded_nodes, uint32_t triangle_count);

int ptr_dispatch(ptr_ctx* ctx, const char* kernel, const uint8_t constants[64], uint32_t groups_x, uint32_t groups_y,
                 uint32_t groups_z);

/* The same "compute" dispatch for callers that cannot allocate device memory -- code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  It has no counterpart in the reference.  ptr_dispatch keeps the queues, hit
 * records, counters and traversal-stack spill area of the wavefront pipeline in scratch it allocates (and re-allocates when a
 * larger frame or a deeper tree comes along); this entry point keeps them in ONE fixed 34 MB array per GPU that is part of the
 * library's device image and stays until the process ends, and calls no hipMalloc / hipFree / stream or event creation on the
 * way, nor does it wait for the stream: what would need a copy back to the host -- scene metadata in a buffer above 4 KB (no
 * host copy is kept of those), a layout header no earlier ptr_dispatch has read -- fails with PTR_E_STATE instead.  Same
 * arguments, same images bit for bit.  Limits (PTR_E_RANGE beyond them): a frame of at most 163 840 queue slots
 * -- ceil(width/8) * (ceil(height/8) + stripe_rows/8) * 64, e.g. 512 x 288 -- and a flattened layout whose header asks for a
 * traversal stack of at most 61 entries.  What ptr_dispatch does once per upload with allocations is not repeated here: the
 * scene buffers must have passed the contents check in an earlier ptr_dispatch (PTR_E_STATE otherwise, unless
 * "scene.validate" is 0), and the per-triangle shading records are used when that earlier dispatch built them (else the shade
 * step gathers per hit, same results).  The frame runs start to end on the context's stream ("frames.in_flight" does not
 * apply; no per-kernel profile events), on a smaller persistent traversal grid than ptr_dispatch uses.  Every call on a GPU
 * shares that GPU's array: calls that can overlap on one GPU (two contexts, or one context moved between streams without
 * a ptr_sync) race.  kernel: "compute", "compute.megakernel" (uses no scratch at all) or "compute.wavefront".  While dispatches
 * are held back under "dispatch.coalesce" the call fails with PTR_E_STATE (running them would allocate): ptr_flush first. */
int ptr_dispatch_static(ptr_ctx* ctx, const char* kernel, const uint8_t constants[64], uint32_t groups_x, uint32_t groups_y,
                        uint32_t groups_z);

/* ---- multi-GPU tile partition (no counterpart in the single-GPU reference): this context renders only
 *      the 8-row-aligned stripes s with s % world == rank, stripe = stripe_rows image rows.
 *      world = 1 restores the whole frame. ---- */
int ptr_set_partition(ptr_ctx* ctx, uint32_t rank, uint32_t world, uint32_t stripe_rows);
/* The exchange of the tile partition (no counterpart in the reference): an all-gather moves one contiguous chunk per rank.
 * pack: the stripes `rank` owns of an RGBA32_SFLOAT image, in order, into `packed` -- ceil(stripes / world) stripes of
 * stripe_rows * width * 16 bytes, rows past the image's end zero.  unpack: `world` such chunks in rank order (what the
 * all-gather delivers), chunk r's stripe j being stripe j * world + r of the frame, back into the image's rows.
 * Stream-ordered on the context's stream like a dispatch. */
int ptr_image_pack_stripes(ptr_ctx* ctx, ptr_handle image, uint32_t rank, uint32_t world, uint32_t stripe_rows, ptr_handle packed);
int ptr_image_unpack_stripes(ptr_ctx* ctx, ptr_handle image, uint32_t world, uint32_t stripe_rows, ptr_handle gathered);
/* ---- multi-GPU sample partition (no counterpart in the reference): frame `sample_index` of this context
 *      draws the global sample (sample_index - 1) * world + rank + 1 over the whole frame and keeps the running
 *      mean of its own samples.  The per-GPU means are combined with the two extra dispatch names
 *      "linearize" (output = sRGB-decoded accumulation image, alpha 1) -> sum over GPUs (RCCL all-reduce) ->
 *      "resolve" (output = sRGB-encoded accumulation.rgb / accumulation.a).  world = 1 turns it off. ---- */
int ptr_set_sample_partition(ptr_ctx* ctx, uint32_t rank, uint32_t world);

/* ---- tuning knobs (no counterpart in the reference); results never depend on them ----
 *      "extend.refill"         idle lanes at which a persistent traversal wave fetches new rays: 8, 16 (default),
 *                              32 or 64 (64 = only when the whole wave is idle)
 *      "extend.inner_reps"     1..3 inner-node steps per inner turn of a traversal wave
 *      "extend.alone_turns"    iterations the last ray of a traversal wave walks alone (nothing left to fetch) before all 64
 *                              lanes finish it together, one pending node or leaf per lane (default 0; 1000000 = never)
 *      "build.traversal_cost"  ptr_bvh_build_flat: cost of entering a node in percent of one triangle test (default 100); a
 *                              subtree of up to 4 triangles becomes one leaf when that is cheaper by the surface-area heuristic
 *      "build.cluster_radius"  ptr_bvh_build_flat: 0 = the binary tree comes from the common prefixes of the triangles' Morton
 *                              keys (radix tree); 1..64 = from locally-ordered clustering: every round each cluster merges with
 *                              the neighbour, at most that many places away in Morton order, with which it makes the smallest
 *                              box, where the choice is mutual.  Default 4: 4-7 % fewer node visits per ray than the radix tree on
 *                              street, interior and atrium scenes for a tenth more build time (2 % more on a regular grid of
 *                              copies of a scene, which the Morton cells already cut exactly); larger radii build worse trees
 *                              on all of them.  Either way the layout is the same and images are those of the oracle.
 *      "extend.leaf_all"       1 (default): a leaf turn tests the lane's whole leaf; 0: one triangle per turn
 *      "extend.leaf_weight"    1, 2 (default), 4: how strongly lanes waiting at a leaf outvote lanes waiting at an
 *                              inner node when the wave picks the step kind of its next iteration
 *      "extend.blocks_per_cu"  persistent grid size of the traversal kernel, 1..64 blocks of 256 threads per CU (default 6: what stays resident at the kernel's register count)
 *      "frames.in_flight"      1..4 (default 2): "compute" dispatches of the wavefront pipeline run on that many internal
 *                              streams in turn, each with its own scratch, so that the tail of one frame overlaps the
 *                              start of the next.  The accumulate step (the only one that touches images) is ordered by
 *                              events after everything enqueued on the context's stream before the dispatch and before
 *                              everything enqueued after it, so images, reads and ptr_sync behave as with one stream.
 *                              A write to a buffer a frame in flight still reads waits for that frame first.
 *                              1 = one stream for everything.  (Measured on MI355X: 2 gives 1.38x the frame rate of 1;
 *                              3 is slower than 2 because a fourth active hardware queue costs more than it hides.)
 *      "profile.extend_only"   1: ptr_profile_enable records events around the traversal kernel only (the events between
 *                              the other kernels cost a few per cent of the frame rate); 0 (default): around every kernel.
 *      "scene.index_words"     3 (default): the index buffer holds the reference's packing, three words per triangle, 24-bit
 *                              vertex id | one byte of the 24-bit material id (src/scene.cpp:140-142, ray.h:20-30);
 *                              4: extension for scenes with more than 2^24 vertices (no counterpart in the reference):
 *                              four plain words per triangle, v0, v1, v2, material id; the buffer must be 16-byte aligned.
 *      "scene.validate"        1 (default): before a "compute" dispatch first uses a set of scene buffers (and again after any
 *                              ptr_buffer_write to them) the library checks on the GPU that every vertex id, material id,
 *                              node link and child reference stays inside the buffer it indexes, and fails the dispatch with
 *                              PTR_E_RANGE otherwise (one synchronisation per upload); 0: trust the caller.
 *      "dispatch.samples"      1 (default) .. 64: a "compute" dispatch of the wavefront pipeline renders that many consecutive
 *                              samples -- sample_index, sample_index + 1, ... of the metadata -- in one pass through the kernels
 *                              (every tile is queued that many times) and accumulates them in order; the output image then
 *                              holds what that many single-sample dispatches, with the caller's image swap after each
 *                              (src/main.cpp:416), would have left in the image written last, bit for bit, and the caller
 *                              advances sample_index by the same number.  An extension of the frame protocol for callers
 *                              that want samples rather than every intermediate frame (progressive accumulation, the
 *                              benchmark): a launch then holds several times the rays, so the drain of each persistent
 *                              launch and the launch boundaries weigh less, and a GPU that owns 1/N of the pixels in the
 *                              tile partition still has a full GPU's worth of rays.  Needs output and accumulation images of
 *                              the same extent; not available with ptr_set_capture, the megakernel or debug_bvh
 *                              (PTR_E_UNSUPPORTED / PTR_E_STATE).
 *      "bvh.auto_flatten"      0 (default) | 1: a "compute" dispatch whose bvh buffer has no flattened layout bound -- the caller
 *                              uploaded the reference's seven buffers and nothing else -- builds one on the GPU from the
 *                              dispatch's index and position buffers (as ptr_bvh_build_flat does), takes the tie-break ranks
 *                              from the threaded tree (ptr_bvh_flat_set_reference_ranks), binds it and runs the wavefront
 *                              pipeline instead of the megakernel: same image, about 50 times the speed on the benchmark scene,
 *                              128 + 80 bytes of device memory per triangle, tens of milliseconds once.  The layout is rebuilt
 *                              when the node, index or position buffer is written again, and replaced by a layout the caller
 *                              binds.  Needs the three buffers addressed from their start and the full 2N-1-node tree; a
 *                              dispatch that does not qualify, or whose tree the builder cannot handle, runs the megakernel
 *                              as before.
 *      "dispatch.coalesce"     1 (default) .. 64: the reference's frame loop as it is -- write sample_index, dispatch, swap the
 *                              two images, once per sample (src/main.cpp:406-416) -- at the speed of "dispatch.samples": the
 *                              context holds back up to that many consecutive "compute" dispatches that continue one
 *                              accumulation chain (same constants and metadata except sample_index + 1 and the two image
 *                              indices swapped) and renders them as ONE dispatch of that many samples.  It runs when the
 *                              count is reached, when a dispatch does not continue the chain, and before any other call that
 *                              reads or changes what it sees (ptr_sync, ptr_flush, image and buffer reads, uploads, writes to
 *                              a buffer it reads, options, partitions, destruction; NOT a write to the metadata buffer, whose
 *                              contents a dispatch takes when it is issued).  Both images then hold exactly what the single
 *                              dispatches would have left: the last result, and the one before it.  What differs: a
 *                              dispatch that is held back has not been enqueued when ptr_dispatch returns (use ptr_flush
 *                              before touching the images by other means), and an error it runs into is reported by the
 *                              call that runs it.  Only for dispatches "dispatch.samples" could render (wavefront pipeline,
 *                              no capture, no counting run, no debug_bvh); all others run at once, after the ones held back.
 *      "shade.pregather"       1 (default): the wavefront shade step reads one 80-byte record per hit triangle (vertex
 *                              normals, uvs, material id, geometry normal) that the library gathers once from the index /
 *                              positions / normals / uvs buffers of the dispatch and re-gathers after any ptr_buffer_write
 *                              to them (80 bytes of device memory per triangle); 0: gather from the four arrays per hit,
 *                              as the shader does (ray.h:20-64).  The results are identical. */
int ptr_set_option(ptr_ctx* ctx, const char* key, int value);

/* ---- instrumentation (no counterpart in the reference) ---- */
/* Per-pixel capture of the first two closest hits of the next "compute" dispatches: 16 bytes per pixel
 * {int32 prim0, float t0, int32 prim1, float t1}, prim = -1 for a miss / ended path.  0 = off. */
int ptr_set_capture(ptr_ctx* ctx, ptr_handle buffer);
/* Traversal counters: when on, "compute" dispatches run counting kernel variants.
 * out[0..4] = rays, node records fetched, triangle records fetched, shaded hits, frames;
 * out[5..6] = wave-level inner-node / leaf iterations of the wavefront traversal (lane utilisation =
 * node records / (64 * inner iterations)); out[7] = lanes without a ray, summed over the inner iterations. */
int ptr_stats_enable(ptr_ctx* ctx, int on);
int ptr_stats_read(ptr_ctx* ctx, uint64_t out[8], int reset);
/* Per-kernel device time from HIP events recorded around every launch on the context's stream. */
int ptr_profile_enable(ptr_ctx* ctx, int on);
int ptr_profile_read(ptr_ctx* ctx, const char* kernel_name, uint64_t* launches, double* total_ms, int reset);
/* Names the profiler knows, '\n'-separated, into dst (size bytes incl. NUL). */
int ptr_profile_names(ptr_ctx* ctx, char* dst, size_t size);

/* Evaluates ONE device function of the shader on `count` argument records (host arrays of 32-bit words; floats
 * as their bit patterns) and returns the result records, for known-answer tests of the device code at function
 * granularity.  Names and record layouts (words in -> words out):
 *   "pcg_hash" 1->2, "rand" 1->2 (shaders/include/rand.h:1-12; out = value, advanced state);
 *   "sin" 1->1, "cos" 1->1, "pow" 2->1 (the GLSL built-ins used by math.h / brdf.h / color_utils.h);
 *   "srgb_to_linear" 3->3, "linear_to_srgb" 3->3 (color_utils.h:1-21);
 *   "hit_aabb" min[3] max[3] origin[3] dir[3] min_t max_t -> hit (ray.h:66-78);
 *   "hit_triangle" p0[3] p1[3] p2[3] origin[3] dir[3] min_t max_t -> hit, t, bary[3], normal[3] (ray.h:137-163);
 *   "hit_sphere" centre[3] radius origin[3] dir[3] min_t max_t -> hit, t (ray.h:105-135);
 *   "rotation_to_z_axis" 3->4, "rotate_point" q[4] v[3] -> 3, "sample_hemisphere" 2->3, "disk_vec" 2->2 (math.h:12-45);
 *   "sample_ggx_vndf" ve[3] ax ay seed -> 3, "brdf_probability" colour[3] metalness view[3] normal[3] -> 1 (brdf.h:13-35,83-99).
 * in_words / out_words must equal the function's record sizes (PTR_E_INVALID otherwise). Synchronous. */
int ptr_device_function(ptr_ctx* ctx, const char* name, const uint32_t* in, uint32_t in_words, uint32_t* out, uint32_t out_words,
                        uint32_t count);

#ifdef __cplusplus
}
#endif
#endif /* PTR_HIP_H_ */
