// Host-side launch interface of the HIP kernels (implemented in ptr_kernels_mega.hip and
// ptr_kernels_wavefront.hip).  Everything is enqueued on the given stream; nothing synchronises.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ptr_device_shading.h"

namespace ptr {

// Top level over placed copies of a flattened BVH (the blob bvh::flatten_instances() emits), as device pointers.
struct InstanceView {
    const float4* nodes;      // 4 x float4 per 64-byte node, boxes in world space; nullptr = no instancing
    const float4* records;    // 4 x float4 per instance: three rows of world-to-object, then (object root, -, -, -)
    int32_t root;
    uint32_t count;
};

// Flattened BVH (the blob bvh::flatten() emits), as device pointers.
struct FlatView {
    const float4* nodes;      // 4 x float4 per 64-byte node
    const float4* triangles;  // 4 x float4 per 64-byte triangle record
    int32_t root;
    uint32_t max_depth;       // stack entries a traversal can need (both levels together when `top` is set)
    InstanceView top;
};

constexpr uint32_t kInstanceTag = 0x40000000u;     // set in ~reference of an instance leaf of the top level
constexpr uint32_t kLeaveInstance = 0x3ffffffeu;   // instance index of the stack entry that marks the way back to world space

// Scratch of the wavefront pipeline, owned by the context, sized for `capacity` paths.
struct WavefrontScratch {
    // ray + path state, three 16-byte records per path in three arrays (every access of a wave is one
    // contiguous 1 KiB run), two generations (read bounce b, write bounce b+1):
    float4* ray_o[2];       // origin.xyz, pixel (bits; gy * width + gx of the dispatch, 0xffffffff = empty slot)
    float4* ray_d[2];       // direction.xyz, RNG state (bits)
    float4* path_t[2];      // throughput.xyz, unused
    float4* hits;           // t, u, v, prim (bits)
    int32_t* hit_instance;  // instance of the hit (two-level scenes only)
    float4* samples;        // per pixel: the sample's colour (w unused)
    uint32_t* counters;     // [0..15]: live paths entering bounce b; [16..31]: fetch cursors of the extend kernel
    int32_t* stack_spill;   // traversal stack overflow area of the persistent extend grid
    uint32_t* fault;        // host-visible word (pinned memory): set to 1 when a traversal ran out of list space
    uint32_t capacity;
    uint32_t extend_blocks; // persistent grid size
    uint32_t spill_entries; // per-lane entries of stack_spill (sized for the deepest tree bound so far)
    uint32_t alone_turns;   // turns a traversal wave's last ray walks alone before all 64 lanes take it over
};

constexpr uint32_t kMaxBounceSlots = 16;   // counters per array; deeper paths reuse the slots cyclically
constexpr uint32_t kExtendThreads = 256;
constexpr uint32_t kStackLds = 16;         // per-lane traversal stack entries kept in LDS
constexpr uint32_t kStackSpill = 48;       // least number of further entries in global memory; grows with the bound tree's stack need
constexpr uint32_t kStackMax = 254;        // deepest stack a flattened layout may ask for (the oracle's restatement holds 256)

// compute.comp as one kernel over the threaded node array (one wave = one 8x8 workgroup).
hipError_t launch_megakernel(hipStream_t s, const SceneView& sv, uint32_t groups_x, uint32_t groups_y, bool stats);
// tonemapping.comp
hipError_t launch_tonemap(hipStream_t s, const SceneMeta* meta, const ImageDesc* images, uint32_t image_count,
                          uint32_t output_index, uint32_t accumulation_index, uint32_t groups_x, uint32_t groups_y);

// Sample-partition exchange helpers over two same-sized RGBA32F images.
hipError_t launch_linearize_resolve(hipStream_t s, const ImageDesc* images, uint32_t output_index, uint32_t accumulation_index,
                                    uint32_t pixels, bool resolve);

// Fills the lookup copy of a row-major RGBA8 image: one 16-byte entry of 2 x 2 texels per texel position, the entries in
// 2 x 2 tiles (ceil(width / 2) * ceil(height / 2) * 64 bytes).
hipError_t launch_footprints_rgba8(hipStream_t s, const void* linear, void* footprints, uint32_t width, uint32_t height);
// RGBA32F texels -> RGBA8 UNORM words (clamp, x 255, round to nearest): the conversion of the reference's present blit.
hipError_t launch_unorm8_pack(hipStream_t s, const void* image, uint32_t texels, void* packed);

// Tile-partition exchange: copies between the rows of an RGBA32F frame and the packed chunks an all-gather moves.
// pack (unpack = false): ranks = 1, first_rank = the owner; unpack: first_rank = 0, ranks = world.
hipError_t launch_stripes_copy(hipStream_t s, float4* image, uint32_t width, uint32_t height, uint32_t stripe_rows, uint32_t world,
                               uint32_t first_rank, uint32_t ranks, float4* packed, bool unpack);

// Wavefront stages.  domain_w/h = pixels of the dispatch that exist (min(groups*8, width/height)).
hipError_t launch_wf_generate(hipStream_t s, const SceneView& sv, const WavefrontScratch& w, uint32_t domain_w, uint32_t domain_h);
// any_hit: the walk of a ray ends at the first triangle (or sphere) it hits; the published hit then says THAT the ray hits
// something, not what or where.  For the last bounce of a path, where nothing else matters (compute.comp:130-136, 241-244: a path
// that is still alive when the loop runs out contributes nothing, whatever it hit).
hipError_t launch_wf_extend(hipStream_t s, const FlatView& fv, const WavefrontScratch& w, uint32_t bounce, unsigned long long* stats,
                            int refill, int leaf_weight, int leaf_all, int inner_reps, bool any_hit = false);
hipError_t launch_wf_shade(hipStream_t s, const SceneView& sv, const WavefrontScratch& w, uint32_t bounce, uint32_t max_bounce,
                           uint32_t min_bounce, uint32_t domain_w, uint32_t domain_h);
// Fills the per-triangle shading records (kShadeRecordVec4 float4 each) from the index / positions / normals / uvs arrays
// of `sv`; vertex ids are clamped to [0, vertex_count).
hipError_t launch_wf_build_shade_records(hipStream_t s, const SceneView& sv, uint32_t triangle_count, uint32_t vertex_count, float4* records);
hipError_t launch_wf_accumulate(hipStream_t s, const SceneView& sv, const WavefrontScratch& w, uint32_t domain_w, uint32_t domain_h);

// GPU construction of the flattened BVH (ptr_kernels_build.hip) from the index / position buffers of `sv` into `blob`
// (flat_build_blob_bytes(n) bytes of device memory).  Synchronous.  *status: 0 ok, < 0 = why the tree could not be built
// (-1 bad arguments, -3 node capacity or quantisation, -5 deeper than the traversal stack, others internal).
size_t flat_build_blob_bytes(uint32_t triangle_count);
// traversal_cost: cost of entering a node relative to one triangle test, for the leaf decision (surface-area heuristic).
// cluster_radius: 0 = binary tree from the Morton keys' prefixes (radix tree); > 0 = from locally-ordered clustering with
// that search radius (status -6: the clustering did not finish within its round limit).
hipError_t build_flat_bvh(hipStream_t s, const SceneView& sv, uint32_t triangle_count, uint32_t vertex_count, void* blob, size_t blob_bytes, float traversal_cost,
                          int cluster_radius, int* status);

// Writes into the triangle records of a flattened layout the ranks of the reference's visiting order: the position of each
// primitive's leaf among the leaves of the threaded node array (include/bvh.hpp:17-22).  Primitives the array does not hold
// keep their rank.  Synchronous.
hipError_t apply_reference_ranks(hipStream_t s, const ThreadedNode* nodes, uint32_t node_count, float4* records, uint32_t record_count, uint32_t triangle_count);

// Contents check of the buffers a "compute" dispatch dereferences (ptr_kernels_validate.hip).  out4 (device, 4 words):
// max vertex id, max material id, violations in the threaded nodes, violations in the flattened layout (and in the top
// level over its copies, fv->top, when there is one).
// sphere_count: records the layout's sphere leaves may name (0: a sphere leaf is a violation).
hipError_t launch_validate_scene(hipStream_t s, const SceneView& sv, uint32_t triangle_count, uint32_t threaded_nodes, const FlatView* fv,
                                 uint32_t flat_nodes, uint32_t flat_triangles, uint32_t top_nodes, uint32_t sphere_count, uint32_t* out4);

// Stack entries a depth-first walk of the 4-wide tree at `nodes` can have pending at once (the figure a layout header
// claims as max_depth), measured level by level.  Synchronous.  *status: 0 ok, -1 bad root, -2 more than `limit`
// (or a cycle), -3 a node reached along several paths.  Child references must have been range-checked before.
hipError_t measure_stack_need(hipStream_t s, const float4* nodes, uint32_t node_count, int32_t root, uint32_t limit, uint32_t* need, int* status);

// Per-function test kernel (ptr_kernels_unit.hip): index of `name` (< 0 if unknown) with its record sizes in 32-bit words.
int unit_function_lookup(const char* name, uint32_t* in_words, uint32_t* out_words);
hipError_t launch_unit_function(hipStream_t s, int fn, const uint32_t* in, uint32_t in_words, uint32_t* out, uint32_t out_words, uint32_t count);

}  // namespace ptr
