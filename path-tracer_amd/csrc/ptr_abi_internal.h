// Internals of the C ABI (include/ptr_hip.h), shared by its translation units:
//   ptr_abi_context.hip    context, streams, buffers, images, samplers, options, statistics, profile, partitions
//   ptr_abi_dispatch.hip   what a "compute" dispatch does: scene checks, shading records, automatic flattening, the launches
//   ptr_abi_coalesce.hip   dispatches held back under "dispatch.coalesce" and how they run
//   ptr_abi_static.hip     the fixed per-GPU scratch array and ptr_dispatch_static
// Not installed; nothing here is part of the ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ptr_hip.h"
#include "ptr_kernels.h"

namespace ptr_abi {

constexpr uint32_t kMaxImages = 4096;
constexpr size_t kShadowLimit = 4096;  // buffers up to this size keep a host copy (the 160-byte metadata)

struct BufferRec {
    void* ptr = nullptr;
    size_t size = 0;
    bool alive = false;
    ptr_handle flat = 0;  // flattened-layout buffer bound to this (threaded node) buffer
    std::vector<uint8_t> shadow;
    bool flat_view_valid = false;  // this buffer holds a flattened layout whose header was parsed
    uint64_t generation = 0;       // context-wide write counter at this buffer's creation / last write
    ptr::FlatView flat_view{};
    uint32_t flat_nodes = 0, flat_triangles = 0;  // counts from the parsed header
    ptr_handle instances = 0;      // on a flattened-layout buffer: the top level over its placed copies, if any
    ptr_handle spheres = 0;        // on a flattened-layout buffer: the sphere records its sphere leaves name, if any
    uint32_t sphere_count = 0;
    // "bvh.auto_flatten": `flat` was built by the library from the dispatch's own buffers, at these write generations of the
    // node / index / position buffers (a write to one of them makes the layout stale); or the build failed at them
    bool flat_is_auto = false, auto_failed = false;
    uint64_t auto_generations[3] = {};
    bool top_view_valid = false;   // this buffer holds such a top level whose header was parsed
    ptr::InstanceView top_view{};
    uint32_t top_nodes = 0, top_depth = 0;
};
struct ImageRec {
    void* ptr = nullptr;
    uint32_t w = 0, h = 0;
    int format = 0;
    uint32_t sampler = 0;
    bool owned = false, alive = false;
    void* footprints = nullptr;   // RGBA8 images the library owns: the 2 x 2 texels of every bilinear lookup as one entry (what the shade step samples)
};
struct SamplerRec { int filter = 0, address = 0; bool alive = false; };

struct ProfileRec {
    uint64_t launches = 0;
    double total_ms = 0.0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};


}  // namespace ptr_abi

using namespace ptr_abi;

struct ptr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    std::string error;
    std::vector<BufferRec> buffers{ 1 };    // index = handle; 0 unused
    std::vector<ImageRec> images{ 1 };      // index = handle = bindless index
    std::vector<SamplerRec> samplers{ 1 };
    ptr::ImageDesc* d_images = nullptr;     // device copy of the bindless image table
    uint32_t part_rank = 0, part_world = 1, part_stripe = 8;
    uint32_t sample_rank = 0, sample_world = 1;
    ptr_handle capture = 0;
    bool stats_on = false;
    unsigned long long* d_stats = nullptr;  // 8 counters
    uint64_t stats_frames = 0;
    bool profile_on = false;
    int opt_profile_extend_only = 0;  // 1: events around the traversal kernel only (fewer events between the other kernels)
    std::map<std::string, ProfileRec> profile;
    std::vector<hipEvent_t> event_pool;
    // Frames of the wavefront pipeline run on auxiliary streams, kMaxFrameSlots at most in flight, each with its own
    // scratch, so that the latency-bound tail of one frame (few rays left in the last bounces, kernel boundaries)
    // overlaps the start of the next.  Only the accumulate step touches images: it is ordered, by events, after
    // everything enqueued on the context's stream before the dispatch and before everything enqueued after it, so
    // images behave as if the whole frame had run on the context's stream.
    static constexpr int kMaxFrameSlots = 4;
    struct FrameSlot {
        hipStream_t stream = nullptr;
        hipEvent_t ready = nullptr;      // recorded on the context's stream at dispatch: the inputs written before it
        hipEvent_t images = nullptr;     // recorded on the context's stream at dispatch: the image work enqueued before it
        hipEvent_t consumed = nullptr;   // recorded on the slot's stream after the frame's accumulate step
        bool in_flight = false;          // `consumed` has been recorded for a frame of this slot
        uint64_t synced_epoch = 0;       // value of write_epoch when this slot's stream last waited for the context's stream
        uint64_t ready_epoch = 0;        // value of write_epoch when `ready` was last recorded
        ptr::WavefrontScratch scratch{};
        void* scratch_block = nullptr;
        ptr_handle reads[10] = {};       // buffers the slot's last frame reads (hazard check of later writes)
    };
    FrameSlot slots[kMaxFrameSlots];
    int next_slot = 0;
    int opt_frames_in_flight = 2;   // 1 = every frame start to end on the context's stream
    int cu_count = 256;
    int opt_extend_inner_reps = 3;  // inner-node steps per inner turn
    int opt_extend_alone_turns = 0;  // turns a wave's last ray walks alone before the wave finishes it together
    int opt_build_traversal_cost = 100;  // GPU tree build: cost of entering a node, in percent of one triangle test
    int opt_build_cluster_radius = 4;    // GPU tree build: 0 = radix tree over the Morton keys, > 0 = locally-ordered clustering with this search radius
    int opt_extend_leaf_all = 1;    // 1: a leaf turn tests every triangle of the leaf, 0: one triangle per turn
    int opt_extend_leaf_weight = 2; // leaf lanes count this many times when a traversal wave picks its next step kind
    int opt_extend_refill = 16;     // idle lanes that trigger a refill of the traversal wave (8, 16, 32, 48, 64)
    int opt_extend_blocks_per_cu = 6;
    int opt_index_words = 3;        // layout of the index buffer: 3 = the reference's packed words, 4 = v0, v1, v2, material
    int opt_dispatch_samples = 1;   // samples one "compute" dispatch of the wavefront pipeline renders (sample_index, +1, ...)
    int opt_shade_pregather = 1;    // wavefront shade reads per-triangle records gathered once per scene
    // "dispatch.coalesce": consecutive single-sample "compute" dispatches that continue one accumulation chain (same scene,
    // sample_index + 1, the two images swapped) are held back and rendered as ONE dispatch of that many samples
    int opt_coalesce = 1;           // dispatches held back at most (1 = every dispatch runs at once)
    int opt_auto_flatten = 0;       // 1: a "compute" dispatch with no flattened layout bound builds one on the GPU and binds it
    int opt_coalesce_scratch_mb = 8192;   // frame scratch a chain of held-back dispatches may ask for (about 132 bytes per pixel and sample)
    int opt_refuse_batches = 0;     // tests only: batched dispatches fail, so that the one-by-one replay of a held-back chain runs
    struct Pending {
        bool active = false;
        uint8_t last[64] = {};      // constant block of the newest dispatch held back
        ptr::SceneMeta meta{};      // metadata of the first one (the others differ in sample_index only)
        uint32_t count = 0, gx = 0, gy = 0;
        int force = 0;
        ptr_handle reads[10] = {};  // buffers the held-back dispatches will read (a write to one of them runs them first)
    };
    Pending pending;
    // the records, and what they were built from: addresses + write generations of indices / positions / normals / uvs
    void* shade_records = nullptr;
    size_t shade_records_bytes = 0;
    uint32_t shade_record_count = 0;
    uint64_t shade_key[8] = {};
    bool shade_key_valid = false;
    uint64_t write_epoch = 0;       // bumped by every buffer creation and write
    // verdicts of the scene-contents check, by what was checked (addresses + write generations); a few entries so that
    // dispatches alternating between scenes or pipelines do not re-check (and synchronise) every time
    struct Validated { uint64_t key[19]; bool used = false; };
    Validated validated[4];
    int validated_next = 0;
    int opt_validate = 1;
    uint32_t* d_validate = nullptr;
    uint64_t shared_epoch = 0;      // write_epoch of the last change to what every frame reads: textures, image table, shading records
    // small host->device writes (the per-frame 160-byte metadata) go through a ring of pinned slots so that
    // Buffer::write returns without draining the stream
    static constexpr int kStageSlots = 32;
    uint8_t* stage = nullptr;              // kStageSlots x kShadowLimit bytes, pinned
    hipEvent_t stage_done[kStageSlots] = {};
    int stage_next = 0;
    // one pinned, device-visible word the traversal kernel sets when it had to give up on a ray (only possible with the
    // scene checks off and a layout header that understates its stack need); reported by the next ptr_sync / ptr_dispatch
    volatile uint32_t* fault = nullptr;
    uint32_t* fault_device = nullptr;
    ptr::WavefrontScratch static_scratch{};   // views into this GPU's fixed scratch array (ptr_dispatch_static)
    void* unorm8_stage = nullptr;             // ptr_image_read_unorm8: the converted texels on the device, grown on demand
    size_t unorm8_stage_bytes = 0;
};

namespace ptr_abi {

// Sets the context's (or, without a context, the creating thread's) error text; returns `code`.
int fail(ptr_ctx* c, int code, const char* fmt, ...);
const char* create_error();

#define PTR_HIP_CHECK(ctx, expr)                                                                       \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(ctx, PTR_E_HIP, "%s: %s", #expr, hipGetErrorString(e_));     \
    } while (0)

// ---- the scratch of ptr_dispatch_static: one fixed array per GPU instead of allocations ----
// Queues, hit records, per-pixel samples, counters and the traversal-stack spill area of ONE frame, laid out like the
// allocated scratch of a frame slot.  Each GPU holds its own copy of the array at its own address; host code must use
// that address (looked up once per device, below), never the array's name.
// (The array itself and the look-up of its address are in ptr_abi_static.hip.)
constexpr uint32_t kStaticPaths = 160u * 1024u;   // queue slots (whole 8x8 tiles of the frame plus one stripe of padding) a call may need
constexpr uint32_t kStaticExtendBlocks = 256u;    // persistent traversal grid of the static variant (256-thread blocks)
constexpr uint32_t kStaticSpill = ptr::kStackSpill;
constexpr uint32_t kStaticMaxDepth = ptr::kStackLds + kStaticSpill - 3u;   // deepest layout the spill area covers
constexpr size_t kStaticSpillWords = (size_t)kStaticExtendBlocks * (ptr::kExtendThreads / 64) * kStaticSpill * 64;
constexpr size_t kStaticWords = (size_t)kStaticPaths * (2 * 12 + 8 + 1) + 64 + kStaticSpillWords;   // 34.2 MB
// With the context's device current: points c->static_scratch into this GPU's copy of the array.
hipError_t bind_static_scratch(ptr_ctx* c);

int bind_device(ptr_ctx* c);
// Every entry point that looks at or changes what a dispatch sees starts here: held-back dispatches run first.
int enter(ptr_ctx* c);
BufferRec* find_buffer(ptr_ctx* c, ptr_handle h);
ImageRec* find_image(ptr_ctx* c, ptr_handle h);
size_t texel_size(int format);
// The registered buffer that contains [addr, addr + need), or null.
BufferRec* buffer_at(ptr_ctx* c, uint64_t addr, size_t need, ptr_handle* handle = nullptr);
int upload_image_desc(ptr_ctx* c, ptr_handle h);
hipEvent_t take_event(ptr_ctx* c);
void drain_profile(ptr_ctx* c);
int drain_frames(ptr_ctx* c);
bool frame_reads(const ptr_ctx* c, ptr_handle h);
int take_fault(ptr_ctx* c);
int ensure_slot(ptr_ctx* c, ptr_ctx::FrameSlot& slot);
int ensure_scratch(ptr_ctx* c, ptr_ctx::FrameSlot& slot, uint32_t capacity, uint32_t stack_need);

// events around a launch, handed to the profile when the scope ends
struct ScopedProfile {
    ptr_ctx* c; const char* name; hipEvent_t start = nullptr, stop = nullptr;
    hipStream_t stream;
    ScopedProfile(ptr_ctx* ctx, const char* n, hipStream_t on = nullptr) : c(ctx), name(n), stream(on ? on : ctx->stream) {
        if (!c->profile_on || (c->opt_profile_extend_only && strcmp(n, "wavefront.extend") != 0)) return;
        start = take_event(c); stop = take_event(c);
        if (start && stop) (void)hipEventRecord(start, stream);
    }
    ~ScopedProfile() {
        if (!c->profile_on || !start || !stop) return;
        (void)hipEventRecord(stop, stream);
        c->profile[name].pending.emplace_back(start, stop);
    }
};

struct ParsedConstants {
    uint64_t addr[7];
    uint32_t output_index, accumulation_index;
};
inline ParsedConstants parse_constants(const uint8_t* k) {
    ParsedConstants p;
    memcpy(p.addr, k, 56);
    memcpy(&p.output_index, k + 56, 4);
    memcpy(&p.accumulation_index, k + 60, 4);
    return p;
}

int read_meta(ptr_ctx* c, uint64_t addr, ptr::SceneMeta* out, bool may_wait = true);
int check_storage_image(ptr_ctx* c, uint32_t index, const char* what);
// no_alloc: the ptr_dispatch_static variant; batch: dispatches that were held back, rendered now as one (ptr_abi_dispatch.hip)
int dispatch_compute(ptr_ctx* c, const uint8_t constants[64], uint32_t gx, uint32_t gy, int force /*0 auto,1 mega,2 wavefront*/, bool no_alloc = false,
                     const ptr_ctx::Pending* batch = nullptr);
int flush_pending(ptr_ctx* c);
int dispatch_or_hold(ptr_ctx* c, const uint8_t constants[64], uint32_t gx, uint32_t gy, int force);

}  // namespace ptr_abi
