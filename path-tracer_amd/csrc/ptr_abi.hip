// C ABI of the render hot path (declared in include/ptr_hip.h): context, device resources, bindless
// tables and the dispatch that turns the reference's 64-byte constant block into HIP launches.
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

namespace {

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

thread_local std::string g_create_error;

}  // namespace

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
        ptr_handle reads[9] = {};        // buffers the slot's last frame reads (hazard check of later writes)
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
        ptr_handle reads[9] = {};   // buffers the held-back dispatches will read (a write to one of them runs them first)
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
    struct Validated { uint64_t key[16]; bool used = false; };
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

namespace {

int fail(ptr_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->error = buf; else g_create_error = buf;
    return code;
}

#define PTR_HIP_CHECK(ctx, expr)                                                                       \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(ctx, PTR_E_HIP, "%s: %s", #expr, hipGetErrorString(e_));     \
    } while (0)

// ---- the scratch of ptr_dispatch_static: one fixed array per GPU instead of allocations ----
// Queues, hit records, per-pixel samples, counters and the traversal-stack spill area of ONE frame, laid out like the
// allocated scratch of a frame slot.  Each GPU holds its own copy of the array at its own address; host code must use
// that address (looked up once per device, below), never the array's name.
constexpr uint32_t kStaticPaths = 160u * 1024u;   // queue slots (whole 8x8 tiles of the frame plus one stripe of padding) a call may need
constexpr uint32_t kStaticExtendBlocks = 256u;    // persistent traversal grid of the static variant (256-thread blocks)
constexpr uint32_t kStaticSpill = ptr::kStackSpill;
constexpr uint32_t kStaticMaxDepth = ptr::kStackLds + kStaticSpill - 3u;   // deepest layout the spill area covers
constexpr size_t kStaticSpillWords = (size_t)kStaticExtendBlocks * (ptr::kExtendThreads / 64) * kStaticSpill * 64;
constexpr size_t kStaticWords = (size_t)kStaticPaths * (2 * 12 + 8 + 1) + 64 + kStaticSpillWords;   // 34.2 MB
static __device__ __align__(256) uint32_t scratch[kStaticWords];

constexpr int kMaxDevices = 64;
std::mutex g_static_mutex;
uint32_t* g_static_scratch[kMaxDevices] = {};   // device address of `scratch` on each GPU that has a context

// With `device` current: the address of this GPU's copy of the array.
hipError_t static_scratch_address(int device, uint32_t** out) {
    std::lock_guard<std::mutex> lock(g_static_mutex);
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    if (!g_static_scratch[device]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        g_static_scratch[device] = (uint32_t*)p;
    }
    *out = g_static_scratch[device];
    return hipSuccess;
}

int bind_device(ptr_ctx* c) {
    PTR_HIP_CHECK(c, hipSetDevice(c->device));
    return PTR_OK;
}
// Every entry point that looks at or changes what a dispatch sees starts here: held-back dispatches run first.
int flush_pending(ptr_ctx* c);
int enter(ptr_ctx* c) {
    int rc = bind_device(c);
    return rc ? rc : flush_pending(c);
}

BufferRec* find_buffer(ptr_ctx* c, ptr_handle h) {
    if (h == 0 || h >= c->buffers.size() || !c->buffers[h].alive) return nullptr;
    return &c->buffers[h];
}
ImageRec* find_image(ptr_ctx* c, ptr_handle h) {
    if (h == 0 || h >= c->images.size() || !c->images[h].alive) return nullptr;
    return &c->images[h];
}
size_t texel_size(int format) { return format == PTR_FORMAT_RGBA8_UNORM ? 4 : (format == PTR_FORMAT_RGBA32_SFLOAT ? 16 : 0); }

// The registered buffer that contains [addr, addr + need), or null.
BufferRec* buffer_at(ptr_ctx* c, uint64_t addr, size_t need, ptr_handle* handle = nullptr) {
    for (size_t i = 1; i < c->buffers.size(); i++) {
        BufferRec& b = c->buffers[i];
        if (!b.alive) continue;
        const uint64_t base = (uint64_t)(uintptr_t)b.ptr;
        if (addr >= base && addr + need <= base + b.size) {
            if (handle) *handle = (ptr_handle)i;
            return &b;
        }
    }
    return nullptr;
}

int upload_image_desc(ptr_ctx* c, ptr_handle h) {
    const ImageRec& r = c->images[h];
    ptr::ImageDesc d;
    d.data = r.alive ? r.ptr : nullptr;
    d.width = r.w; d.height = r.h; d.format = (uint32_t)r.format; d.sampler = r.sampler;
    d.footprints = r.alive ? (const uint32_t*)r.footprints : nullptr;
    PTR_HIP_CHECK(c, hipMemcpyAsync(c->d_images + h, &d, sizeof d, hipMemcpyHostToDevice, c->stream));
    // the descriptor lives on the stack: finish the copy before returning
    PTR_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    c->shared_epoch = ++c->write_epoch;
    return PTR_OK;
}

// ---- profiling: events around a launch ----
hipEvent_t take_event(ptr_ctx* c) {
    if (!c->event_pool.empty()) { hipEvent_t e = c->event_pool.back(); c->event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
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
void drain_profile(ptr_ctx* c) {
    for (auto& kv : c->profile) {
        for (auto& p : kv.second.pending) {
            float ms = 0.f;
            if (hipEventSynchronize(p.second) == hipSuccess && hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
                kv.second.launches++;
                kv.second.total_ms += ms;
            }
            c->event_pool.push_back(p.first);
            c->event_pool.push_back(p.second);
        }
        kv.second.pending.clear();
    }
}

// ---- frames in flight ----
// Host-blocking: every frame that was dispatched has finished.  Taken before anything a frame in flight could still be
// reading is changed or freed.
int drain_frames(ptr_ctx* c) {
    for (auto& slot : c->slots)
        if (slot.stream && slot.in_flight) PTR_HIP_CHECK(c, hipStreamSynchronize(slot.stream));
    return PTR_OK;
}
bool frame_reads(const ptr_ctx* c, ptr_handle h) {
    for (const auto& slot : c->slots)
        if (slot.in_flight)
            for (ptr_handle r : slot.reads) if (r == h && h != 0) return true;
    return false;
}
// A traversal kernel that had to give up on a ray says so in the context's host-visible fault word.
int take_fault(ptr_ctx* c) {
    if (c->fault && *c->fault != 0u) {
        *c->fault = 0u;
        return fail(c, PTR_E_RANGE, "an earlier frame's traversal ran out of list space (the layout header understates the tree's stack need): that frame's image is incomplete");
    }
    return PTR_OK;
}
int ensure_slot(ptr_ctx* c, ptr_ctx::FrameSlot& slot) {
    if (slot.stream) return PTR_OK;
    // (Plain priority on purpose: frame streams in the high-priority pool would get hardware queues of their own, but
    // their event waits on the normal-priority context stream were seen to hang the device.)
    PTR_HIP_CHECK(c, hipStreamCreateWithFlags(&slot.stream, hipStreamNonBlocking));
    PTR_HIP_CHECK(c, hipEventCreateWithFlags(&slot.ready, hipEventDisableTiming));
    PTR_HIP_CHECK(c, hipEventCreateWithFlags(&slot.images, hipEventDisableTiming));
    PTR_HIP_CHECK(c, hipEventCreateWithFlags(&slot.consumed, hipEventDisableTiming));
    return PTR_OK;
}

// ---- wavefront scratch (one per frame slot) ----
int ensure_scratch(ptr_ctx* c, ptr_ctx::FrameSlot& slot, uint32_t capacity, uint32_t stack_need) {
    // spill entries per lane: what the tree can need beyond the LDS part, plus the three pushes of the step in progress
    uint32_t spill = stack_need + 3u > ptr::kStackLds ? stack_need + 3u - ptr::kStackLds : 0u;
    spill = spill < ptr::kStackSpill ? ptr::kStackSpill : ((spill + 15u) & ~15u);
    if (slot.scratch.capacity >= capacity && slot.scratch_block && slot.scratch.spill_entries >= spill &&
        slot.scratch.extend_blocks == (uint32_t)c->cu_count * (uint32_t)c->opt_extend_blocks_per_cu) return PTR_OK;
    if (slot.scratch.capacity > capacity) capacity = slot.scratch.capacity;          // never shrink what another scene still needs
    if (slot.scratch.spill_entries > spill) spill = slot.scratch.spill_entries;
    if (slot.scratch_block) {
        int rc = drain_frames(c); if (rc) return rc;
        PTR_HIP_CHECK(c, hipStreamSynchronize(c->stream));
        PTR_HIP_CHECK(c, hipFree(slot.scratch_block));
        slot.scratch_block = nullptr;
    }
    const size_t cap = ((size_t)capacity + 255) & ~(size_t)255;
    const uint32_t extend_blocks = (uint32_t)c->cu_count * (uint32_t)c->opt_extend_blocks_per_cu;
    const size_t spill_words = (size_t)extend_blocks * (ptr::kExtendThreads / 64) * spill * 64;
    // 2 generations x 3 records of 4 words, hits + samples (4 words each), the hit's instance, counters, spill
    const size_t words = cap * (2 * 12 + 8 + 1) + 64 + spill_words;
    PTR_HIP_CHECK(c, hipMalloc(&slot.scratch_block, words * 4));
    uint32_t* p = (uint32_t*)slot.scratch_block;
    auto take = [&](size_t n) { uint32_t* r = p; p += n; return r; };
    ptr::WavefrontScratch& s = slot.scratch;
    s.hits = (float4*)take(cap * 4);
    s.samples = (float4*)take(cap * 4);
    for (int g = 0; g < 2; g++) {
        s.ray_o[g] = (float4*)take(cap * 4);
        s.ray_d[g] = (float4*)take(cap * 4);
        s.path_t[g] = (float4*)take(cap * 4);
    }
    s.hit_instance = (int32_t*)take(cap);
    s.counters = take(64);
    s.stack_spill = (int32_t*)take(spill_words);
    s.capacity = (uint32_t)cap;
    s.extend_blocks = extend_blocks;
    s.spill_entries = spill;
    return PTR_OK;
}

struct ParsedConstants {
    uint64_t addr[7];
    uint32_t output_index, accumulation_index;
};
ParsedConstants parse_constants(const uint8_t* k) {
    ParsedConstants p;
    memcpy(p.addr, k, 56);
    memcpy(&p.output_index, k + 56, 4);
    memcpy(&p.accumulation_index, k + 60, 4);
    return p;
}

// Host view of the 160-byte metadata the constant block points at.
// may_wait = false (ptr_dispatch_static): a metadata buffer without a host copy is refused instead of being copied back, which
// would make the caller wait for the stream
int read_meta(ptr_ctx* c, uint64_t addr, ptr::SceneMeta* out, bool may_wait = true) {
    ptr_handle h = 0;
    BufferRec* b = buffer_at(c, addr, sizeof(ptr::SceneMeta), &h);
    if (!b) return fail(c, PTR_E_STATE, "constant block: scene metadata address is not inside a registered buffer");
    const size_t off = (size_t)(addr - (uint64_t)(uintptr_t)b->ptr);
    if (!b->shadow.empty()) { memcpy(out, b->shadow.data() + off, sizeof *out); return PTR_OK; }
    if (!may_wait) return fail(c, PTR_E_STATE, "ptr_dispatch_static: the scene metadata lies in a buffer larger than %zu bytes, of which the library keeps no host copy (reading it back would wait for the stream)", kShadowLimit);
    PTR_HIP_CHECK(c, hipMemcpyAsync(out, (const uint8_t*)b->ptr + off, sizeof *out, hipMemcpyDeviceToHost, c->stream));
    PTR_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return PTR_OK;
}

int check_storage_image(ptr_ctx* c, uint32_t index, const char* what) {
    ImageRec* im = find_image(c, index);
    if (!im) return fail(c, PTR_E_STATE, "constant block: %s image index %u is not a live image", what, index);
    if (im->format != PTR_FORMAT_RGBA32_SFLOAT) return fail(c, PTR_E_STATE, "constant block: %s image must be RGBA32_SFLOAT", what);
    return PTR_OK;
}

// Per-triangle shading records of the wavefront shade step: (re)built when the index / positions / normals / uvs
// addresses of the dispatch, or the contents of the buffers behind them, differ from what the records were built
// from.  On success sv names the records; with the option off, or an empty index buffer, sv keeps the gather path.
int ensure_shade_records(ptr_ctx* c, const uint64_t addr[7], const ptr_handle handles[7], ptr::SceneView* sv, bool lookup_only = false) {
    sv->shade_records = nullptr;
    sv->shade_record_count = 0;
    if (!c->opt_shade_pregather) return PTR_OK;
    auto bytes_from = [&](int i) {
        const BufferRec& b = c->buffers[handles[i]];
        return (uint64_t)(uintptr_t)b.ptr + b.size - addr[i];
    };
    const uint64_t tris = bytes_from(2) / (4u * (uint64_t)c->opt_index_words);
    uint64_t vertices = bytes_from(3) / 12u;
    if (bytes_from(4) / 12u < vertices) vertices = bytes_from(4) / 12u;
    if (bytes_from(5) / 8u < vertices) vertices = bytes_from(5) / 8u;
    if (tris == 0 || vertices == 0 || tris > 0x7fffffffull) return PTR_OK;
    if (c->opt_index_words == 3 && vertices > (1u << 24)) vertices = 1u << 24;  // the packed index words hold 24-bit vertex ids
    if (vertices > 0xffffffffull) vertices = 0xffffffffull;
    uint64_t key[8];
    for (int i = 0; i < 4; i++) { key[i] = addr[2 + i]; key[4 + i] = c->buffers[handles[2 + i]].generation; }
    key[0] ^= (uint64_t)c->opt_index_words << 60;  // device addresses leave the top bits free
    if (c->shade_key_valid && c->shade_record_count == tris && !memcmp(key, c->shade_key, sizeof key)) {
        sv->shade_records = (const float4*)c->shade_records;
        sv->shade_record_count = c->shade_record_count;
        return PTR_OK;
    }
    if (lookup_only) return PTR_OK;   // no records for these buffers yet: the shade step gathers per hit (same results)
    const size_t need = (size_t)tris * ptr::kShadeRecordVec4 * sizeof(float4);
    c->shade_key_valid = false;
    { int rc = drain_frames(c); if (rc) return rc; }  // frames in flight read the records that are about to change
    if (need != c->shade_records_bytes) {
        if (c->shade_records) {
            PTR_HIP_CHECK(c, hipStreamSynchronize(c->stream));  // frames in flight may still read the old records
            PTR_HIP_CHECK(c, hipFree(c->shade_records));
            c->shade_records = nullptr; c->shade_records_bytes = 0;
        }
        PTR_HIP_CHECK(c, hipMalloc(&c->shade_records, need));
        c->shade_records_bytes = need;
    }
    {
        ScopedProfile prof(c, "wavefront.build_shade_records");
        PTR_HIP_CHECK(c, ptr::launch_wf_build_shade_records(c->stream, *sv, (uint32_t)tris, (uint32_t)vertices, (float4*)c->shade_records));
    }
    c->shared_epoch = ++c->write_epoch;
    memcpy(c->shade_key, key, sizeof key);
    c->shade_key_valid = true;
    c->shade_record_count = (uint32_t)tris;
    sv->shade_records = (const float4*)c->shade_records;
    sv->shade_record_count = c->shade_record_count;
    return PTR_OK;
}

// The index words and the tree layouts hold indices into other buffers: checked on the GPU once per upload, so that a
// bad scene fails the dispatch instead of faulting the device.
int validate_scene(ptr_ctx* c, const uint64_t addr[7], const ptr_handle handles[7], const ptr::SceneView& sv, BufferRec* flat_buf, bool wavefront, bool lookup_only = false) {
    if (!c->opt_validate) return PTR_OK;
    uint64_t key[16] = {};
    for (int i = 1; i < 7; i++) { key[i - 1] = addr[i]; key[5 + i] = c->buffers[handles[i]].generation; }
    key[12] = flat_buf ? flat_buf->generation : 0;
    key[13] = (uint64_t)c->opt_index_words | (wavefront ? 16u : 0u);
    BufferRec* top_buf = wavefront && flat_buf && flat_buf->instances ? find_buffer(c, flat_buf->instances) : nullptr;
    key[14] = top_buf ? top_buf->generation : 0;
    key[15] = top_buf ? (uint64_t)flat_buf->instances : 0;
    for (const auto& v : c->validated) if (v.used && !memcmp(v.key, key, sizeof key)) return PTR_OK;
    if (lookup_only)
        return fail(c, PTR_E_STATE, "ptr_dispatch_static: these scene buffers have not been checked since their last write; the check allocates, "
                                    "so run one ptr_dispatch on them first (or set scene.validate to 0)");

    auto bytes_from = [&](int i) { const BufferRec& b = c->buffers[handles[i]]; return (uint64_t)(uintptr_t)b.ptr + b.size - addr[i]; };
    const uint64_t triangles = bytes_from(2) / (4u * (uint64_t)c->opt_index_words);
    uint64_t vertices = bytes_from(3) / 12u;
    if (bytes_from(4) / 12u < vertices) vertices = bytes_from(4) / 12u;
    if (bytes_from(5) / 8u < vertices) vertices = bytes_from(5) / 8u;
    const uint64_t materials = bytes_from(6) / sizeof(ptr::Material);
    // the reference's builder emits 2N-1 nodes for N triangles (src/bvh.cpp:48-86); a larger buffer's tail is not part of the tree
    uint64_t threaded = wavefront ? 0u : bytes_from(1) / sizeof(ptr::ThreadedNode);
    if (triangles && threaded > 2 * triangles - 1) threaded = 2 * triangles - 1;
    if (triangles > 0xffffffffull || threaded > 0xffffffffull) return fail(c, PTR_E_RANGE, "scene: buffers larger than 2^32 records");
    if (!c->d_validate) PTR_HIP_CHECK(c, hipMalloc((void**)&c->d_validate, 16));
    ptr::FlatView view{};
    const ptr::FlatView* fv = nullptr;
    if (wavefront && flat_buf) {
        view = flat_buf->flat_view;
        view.top = top_buf ? top_buf->top_view : ptr::InstanceView{};
        fv = &view;
    }
    PTR_HIP_CHECK(c, ptr::launch_validate_scene(c->stream, sv, (uint32_t)triangles, (uint32_t)threaded, fv, fv ? flat_buf->flat_nodes : 0u,
                                                fv ? flat_buf->flat_triangles : 0u, top_buf ? top_buf->top_nodes : 0u, c->d_validate));
    uint32_t out[4];
    PTR_HIP_CHECK(c, hipMemcpyAsync(out, c->d_validate, sizeof out, hipMemcpyDeviceToHost, c->stream));
    PTR_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    if (triangles && out[0] >= vertices) return fail(c, PTR_E_RANGE, "scene: the index buffer names vertex %u, the vertex buffers hold %llu", out[0], (unsigned long long)vertices);
    if (triangles && out[1] >= materials) return fail(c, PTR_E_RANGE, "scene: the index buffer names material %u, the material buffer holds %llu", out[1], (unsigned long long)materials);
    if (out[2]) return fail(c, PTR_E_RANGE, "scene: %u links or primitive ids of the bvh nodes point outside the node / index buffers", out[2]);
    if (out[3]) return fail(c, PTR_E_RANGE, "scene: %u child references or primitive ids of the flattened layout%s point outside it", out[3], top_buf ? " or of its instances" : "");
    if (fv) {
        // the kernels size their traversal stacks from the header's bound: measure what the tree really needs
        uint32_t need = 0, top_need = 0;
        int status = 0;
        PTR_HIP_CHECK(c, ptr::measure_stack_need(c->stream, fv->nodes, flat_buf->flat_nodes, fv->root, ptr::kStackMax, &need, &status));
        if (status == -3) return fail(c, PTR_E_RANGE, "scene: the flattened layout is not a tree (a node is reached along several paths)");
        if (status != 0 || need > flat_buf->flat_view.max_depth)
            return fail(c, PTR_E_RANGE, "scene: the flattened layout needs a deeper traversal stack than its header states (%u)", flat_buf->flat_view.max_depth);
        if (top_buf) {
            PTR_HIP_CHECK(c, ptr::measure_stack_need(c->stream, fv->top.nodes, top_buf->top_nodes, fv->top.root, ptr::kStackMax, &top_need, &status));
            if (status == -3) return fail(c, PTR_E_RANGE, "scene: the top level over the instances is not a tree");
            if (status != 0 || top_need + 1u + flat_buf->flat_view.max_depth > top_buf->top_depth)
                return fail(c, PTR_E_RANGE, "scene: the instances need a deeper traversal stack than their header states (%u)", top_buf->top_depth);
        }
    }
    ptr_ctx::Validated& slot = c->validated[c->validated_next];
    c->validated_next = (c->validated_next + 1) % 4;
    memcpy(slot.key, key, sizeof key);
    slot.used = true;
    return PTR_OK;
}

// "bvh.auto_flatten": the strict drop-in caller uploads the reference's buffers and nothing else; its "compute" dispatches would
// walk the reference's threaded tree in the megakernel (hundreds of nodes per ray).  With the option set the library builds the
// flattened layout itself -- GPU builder over the dispatch's index and position buffers, tie-break ranks taken from the threaded
// tree so that equal-t hits resolve as the reference's visiting order does -- binds it to the node buffer, and the dispatch takes
// the wavefront pipeline.  Rebuilt when one of the three buffers is written again.  Returns PTR_OK with *flat_out = 0 when the
// dispatch does not qualify (buffers not addressed from their start, no full threaded tree, build failed): the megakernel runs.
int auto_flatten(ptr_ctx* c, const uint64_t addr[7], const ptr_handle handles[7], ptr_handle* flat_out) {
    *flat_out = 0;
    const uint64_t gens[3] = { c->buffers[handles[1]].generation, c->buffers[handles[2]].generation, c->buffers[handles[3]].generation };
    {
        BufferRec& nb = c->buffers[handles[1]];
        if (nb.flat && nb.flat_is_auto && !memcmp(nb.auto_generations, gens, sizeof gens)) { *flat_out = nb.flat; return PTR_OK; }
        if (nb.flat && !nb.flat_is_auto) { *flat_out = nb.flat; return PTR_OK; }
        if (nb.auto_failed && !memcmp(nb.auto_generations, gens, sizeof gens)) return PTR_OK;
    }
    for (int i = 1; i <= 3; i++)
        if (addr[i] != (uint64_t)(uintptr_t)c->buffers[handles[i]].ptr) return PTR_OK;
    const uint64_t triangles = c->buffers[handles[2]].size / (4u * (uint64_t)c->opt_index_words);
    if (triangles == 0 || triangles >= (1u << 29)) return PTR_OK;
    if (c->buffers[handles[1]].size / sizeof(ptr::ThreadedNode) < 2 * triangles - 1) return PTR_OK;   // no full tree to take the ranks from
    int rc = drain_frames(c);
    if (rc) return rc;
    if (c->buffers[handles[1]].flat_is_auto && c->buffers[handles[1]].flat) {   // the stale layout goes
        const ptr_handle old = c->buffers[handles[1]].flat;
        PTR_HIP_CHECK(c, hipStreamSynchronize(c->stream));
        PTR_HIP_CHECK(c, hipFree(c->buffers[old].ptr));
        c->buffers[old] = BufferRec();
        c->buffers[handles[1]].flat = 0;
    }
    ptr_handle flat = 0;
    rc = ptr_bvh_build_flat(c, handles[2], handles[3], (uint32_t)triangles, &flat);
    if (rc == PTR_OK) rc = ptr_bvh_flat_set_reference_ranks(c, flat, handles[1], (uint32_t)triangles);
    BufferRec& nb = c->buffers[handles[1]];   // (the buffer table may have grown: look the record up again)
    memcpy(nb.auto_generations, gens, sizeof gens);
    if (rc != PTR_OK) {
        // e.g. a tree deeper than the traversal stack: the dispatch is still answered, by the kernel that needs no layout
        if (flat) { (void)hipFree(c->buffers[flat].ptr); c->buffers[flat] = BufferRec(); }
        nb.auto_failed = true; nb.flat_is_auto = false; nb.flat = 0;
        return PTR_OK;
    }
    nb.flat = flat; nb.flat_is_auto = true; nb.auto_failed = false;
    *flat_out = flat;
    return PTR_OK;
}

// no_alloc: the ptr_dispatch_static variant -- the frame runs start to end on the context's stream in this GPU's fixed
// scratch array, and nothing on the way allocates (no scene check, no record gathering, no streams or events).
// batch: dispatches that were held back ("dispatch.coalesce"), rendered now as one dispatch of batch->count samples; the
// constants are the newest one's, the metadata the first one's.
int dispatch_compute(ptr_ctx* c, const uint8_t constants[64], uint32_t gx, uint32_t gy, int force /*0 auto,1 mega,2 wavefront*/, bool no_alloc = false,
                     const ptr_ctx::Pending* batch = nullptr) {
    const ParsedConstants k = parse_constants(constants);
    ptr::SceneMeta meta;
    int rc = PTR_OK;
    if (batch) meta = batch->meta;
    else if ((rc = read_meta(c, k.addr[0], &meta, !no_alloc))) return rc;
    if (meta.width == 0 || meta.height == 0) return fail(c, PTR_E_STATE, "scene metadata: zero width/height");
    if (meta.downscale_factor < 1) return fail(c, PTR_E_STATE, "scene metadata: downscale_factor < 1");
    if (meta.sample_index == 0) return fail(c, PTR_E_STATE, "scene metadata: sample_index must start at 1 (1/sample_index weights the frame)");
    // pixel and queue-slot numbers are 32-bit products of these two
    if (meta.width > 65535u || meta.height > 65535u || (uint64_t)meta.width * meta.height > (1ull << 28))
        return fail(c, PTR_E_RANGE, "scene metadata: %u x %u is more than the path addresses (65535 per side, 2^28 pixels)", meta.width, meta.height);
    if ((rc = take_fault(c))) return rc;

    static const char* names[7] = { "scene", "bvh", "indices", "positions", "normals", "uvs", "materials" };
    ptr_handle handles[7] = {};
    for (int i = 1; i < 7; i++) {
        if (!buffer_at(c, k.addr[i], 1, &handles[i])) return fail(c, PTR_E_STATE, "constant block: %s address is not inside a registered buffer", names[i]);
    }
    if ((rc = check_storage_image(c, k.output_index, "output"))) return rc;
    if ((rc = check_storage_image(c, k.accumulation_index, "accumulation"))) return rc;

    ptr::SceneView sv{};
    sv.meta = meta;
    sv.nodes = (const ptr::ThreadedNode*)(uintptr_t)k.addr[1];
    sv.indices = (const uint32_t*)(uintptr_t)k.addr[2];
    sv.index_words = (uint32_t)c->opt_index_words;
    if (c->opt_index_words == 4 && (k.addr[2] & 15u)) return fail(c, PTR_E_STATE, "constant block: a 4-word index buffer must be 16-byte aligned");
    sv.positions = (const float*)(uintptr_t)k.addr[3];
    sv.normals = (const float*)(uintptr_t)k.addr[4];
    sv.uvs = (const float*)(uintptr_t)k.addr[5];
    sv.materials = (const ptr::Material*)(uintptr_t)k.addr[6];
    sv.images = c->d_images;
    sv.image_count = (uint32_t)c->images.size();
    sv.output_index = k.output_index;
    sv.accumulation_index = k.accumulation_index;
    sv.part_rank = c->part_rank; sv.part_world = c->part_world; sv.part_stripe_rows = c->part_stripe;
    sv.sample_rank = c->sample_rank; sv.sample_world = c->sample_world;
    sv.capture = nullptr;
    if (c->capture) {
        BufferRec* cb = find_buffer(c, c->capture);
        if (!cb || cb->size < (size_t)meta.width * meta.height * sizeof(ptr::Capture)) return fail(c, PTR_E_STATE, "capture buffer smaller than width*height*16 bytes");
        sv.capture = (ptr::Capture*)cb->ptr;
    }
    sv.stats = c->stats_on ? c->d_stats : nullptr;
    sv.batch_samples = (uint32_t)c->opt_dispatch_samples;
    if (batch && batch->count > 1u && c->opt_refuse_batches) return fail(c, PTR_E_STATE, "batched dispatch refused (debug.refuse_batches)");
    if (batch) {
        // dispatch i read the image dispatch i - 1 wrote: the chain starts from the first one's accumulation image -- the newest
        // one's accumulation image for an odd count, its output image for an even one -- and ends in the newest one's output
        // image; the newest one's accumulation image is left as the dispatch before it wrote it
        sv.batch_samples = batch->count;
        sv.accumulation_index = (batch->count & 1u) ? k.accumulation_index : k.output_index;
        sv.prev_image_plus_one = batch->count >= 2u ? k.accumulation_index + 1u : 0u;
    }

    if (c->opt_auto_flatten && force == 0 && !no_alloc && meta.debug_bvh != 1u) {
        ptr_handle built = 0;
        if ((rc = auto_flatten(c, k.addr, handles, &built))) return rc;
    }
    BufferRec* bvh_buf = &c->buffers[handles[1]];
    BufferRec* flat_buf = bvh_buf->flat ? find_buffer(c, bvh_buf->flat) : nullptr;
    bool wavefront = force == 2 || (force == 0 && flat_buf != nullptr);
    if (meta.debug_bvh == 1u && wavefront) {
        // compute.comp:274-275: the heat-map counts the boxes of the THREADED tree a primary ray enters (ray.h:80-103); it is
        // defined on that tree only, so the frame goes to the kernel that walks it, whatever layout is bound beside it
        const uint64_t triangles = ((uint64_t)(uintptr_t)c->buffers[handles[2]].ptr + c->buffers[handles[2]].size - k.addr[2]) / (4u * (uint64_t)c->opt_index_words);
        const uint64_t node_bytes = (uint64_t)(uintptr_t)bvh_buf->ptr + bvh_buf->size - k.addr[1];
        if (triangles == 0 || node_bytes / sizeof(ptr::ThreadedNode) < 2 * triangles - 1)
            return fail(c, PTR_E_UNSUPPORTED, "debug_bvh: the heat-map counts boxes of the threaded bvh nodes (2N-1 records for N triangles); the bvh buffer of this dispatch holds no such tree");
        if (flat_buf && flat_buf->instances)
            return fail(c, PTR_E_UNSUPPORTED, "debug_bvh: not defined for placed copies (the threaded tree has no instancing)");
        wavefront = false;
    }
    if (sv.batch_samples > 1u) {
        // several samples per dispatch exist in the wavefront pipeline only; their accumulate chain runs in registers
        if (!wavefront) return fail(c, PTR_E_UNSUPPORTED, "dispatch.samples > 1: only the wavefront pipeline renders several samples per dispatch (the megakernel and the debug_bvh heat-map do one)");
        if (sv.capture) return fail(c, PTR_E_STATE, "dispatch.samples > 1: the per-pixel capture holds one sample's hits; set dispatch.samples to 1");
        const ImageRec& oi = c->images[k.output_index];
        const ImageRec& ai = c->images[k.accumulation_index];
        if (oi.w != ai.w || oi.h != ai.h) return fail(c, PTR_E_STATE, "dispatch.samples > 1: the output and accumulation images must have the same extent");
        if ((uint64_t)meta.width * meta.height * sv.batch_samples >= (1ull << 31)) return fail(c, PTR_E_RANGE, "dispatch.samples x width x height must stay below 2^31");
        if (meta.sample_index > 0xffffffffu - sv.batch_samples) return fail(c, PTR_E_RANGE, "sample_index + dispatch.samples overflows");
    }
    if (c->stats_on) c->stats_frames += sv.batch_samples;

    if (!wavefront) {
        // the threaded walk starts at node 0: at least one node must be there
        if (k.addr[1] + 32 > (uint64_t)(uintptr_t)bvh_buf->ptr + bvh_buf->size) return fail(c, PTR_E_STATE, "bvh buffer holds no node");
        if ((rc = validate_scene(c, k.addr, handles, sv, nullptr, false, no_alloc))) return rc;
        if (no_alloc) {   // no events either
            PTR_HIP_CHECK(c, ptr::launch_megakernel(c->stream, sv, gx, gy, c->stats_on));
            return PTR_OK;
        }
        ScopedProfile prof(c, "compute.megakernel");
        PTR_HIP_CHECK(c, ptr::launch_megakernel(c->stream, sv, gx, gy, c->stats_on));
        return PTR_OK;
    }

    if (!flat_buf) return fail(c, PTR_E_STATE, "compute.wavefront: no flattened layout bound to the bvh buffer (ptr_bvh_bind_flat)");
    if (no_alloc && (!flat_buf->flat_view_valid || (flat_buf->instances && !(find_buffer(c, flat_buf->instances) && find_buffer(c, flat_buf->instances)->top_view_valid))))
        return fail(c, PTR_E_STATE, "ptr_dispatch_static: the header of the flattened layout has not been read yet (an earlier ptr_dispatch does that once per upload; here it would wait for the stream)");
    if (!flat_buf->flat_view_valid) {  // parse + validate the header once per upload
        struct Header { uint32_t magic, version, node_count, triangle_count; uint64_t nodes_offset, triangles_offset; int32_t root; uint32_t max_depth; float bounds[6]; } hd;
        if (flat_buf->size < sizeof hd) return fail(c, PTR_E_STATE, "flattened layout: blob shorter than its header");
        PTR_HIP_CHECK(c, hipMemcpyAsync(&hd, flat_buf->ptr, sizeof hd, hipMemcpyDeviceToHost, c->stream));
        PTR_HIP_CHECK(c, hipStreamSynchronize(c->stream));
        if (hd.magic != 0x46525450u || hd.version != 2u) return fail(c, PTR_E_STATE, "flattened layout: bad magic/version");
        if ((hd.nodes_offset & 63u) || (hd.triangles_offset & 63u)) return fail(c, PTR_E_STATE, "flattened layout: sections are not 64-byte aligned");
        if (hd.nodes_offset + (uint64_t)hd.node_count * 64 > flat_buf->size || hd.triangles_offset + (uint64_t)hd.triangle_count * 64 > flat_buf->size)
            return fail(c, PTR_E_STATE, "flattened layout: sections exceed the buffer");
        if (hd.max_depth > ptr::kStackMax) return fail(c, PTR_E_STATE, "flattened layout: needs a traversal stack of %u entries, %u at most", hd.max_depth, ptr::kStackMax);
        if (hd.root >= 0 ? (uint32_t)hd.root >= hd.node_count : ((uint32_t)~hd.root >> 2) + ((uint32_t)~hd.root & 3u) + 1u > hd.triangle_count)
            return fail(c, PTR_E_STATE, "flattened layout: root reference out of range");
        flat_buf->flat_view.nodes = (const float4*)((const uint8_t*)flat_buf->ptr + hd.nodes_offset);
        flat_buf->flat_view.triangles = (const float4*)((const uint8_t*)flat_buf->ptr + hd.triangles_offset);
        flat_buf->flat_view.root = hd.root;
        flat_buf->flat_view.max_depth = hd.max_depth;
        flat_buf->flat_nodes = hd.node_count;
        flat_buf->flat_triangles = hd.triangle_count;
        flat_buf->flat_view_valid = true;
    }
    ptr::FlatView fv = flat_buf->flat_view;
    BufferRec* top_buf = nullptr;
    if (flat_buf->instances) {  // placed copies of this layout: the top level is walked first
        top_buf = find_buffer(c, flat_buf->instances);
        if (!top_buf) return fail(c, PTR_E_STATE, "instances: the buffer bound with ptr_bvh_bind_instances is gone");
        if (!top_buf->top_view_valid) {
            struct Header { uint32_t magic, version, node_count, instance_count; uint64_t nodes_offset, records_offset; int32_t root; uint32_t max_depth; float bounds[6]; } hd;
            if (top_buf->size < sizeof hd) return fail(c, PTR_E_STATE, "instances: blob shorter than its header");
            PTR_HIP_CHECK(c, hipMemcpyAsync(&hd, top_buf->ptr, sizeof hd, hipMemcpyDeviceToHost, c->stream));
            PTR_HIP_CHECK(c, hipStreamSynchronize(c->stream));
            if (hd.magic != 0x49525450u || hd.version != 2u) return fail(c, PTR_E_STATE, "instances: bad magic/version");
            if ((hd.nodes_offset & 63u) || (hd.records_offset & 63u)) return fail(c, PTR_E_STATE, "instances: sections are not 64-byte aligned");
            if (hd.nodes_offset + (uint64_t)hd.node_count * 64 > top_buf->size || hd.records_offset + (uint64_t)hd.instance_count * 64 > top_buf->size)
                return fail(c, PTR_E_STATE, "instances: sections exceed the buffer");
            if (hd.instance_count == 0 || hd.instance_count >= 0x3ffffff0u) return fail(c, PTR_E_STATE, "instances: instance count out of range");
            if (hd.max_depth > ptr::kStackMax) return fail(c, PTR_E_STATE, "instances: needs a traversal stack of %u entries, %u at most", hd.max_depth, ptr::kStackMax);
            if (hd.root >= 0 ? (uint32_t)hd.root >= hd.node_count
                             : (((uint32_t)~hd.root & ptr::kInstanceTag) == 0u || ((uint32_t)~hd.root & ~ptr::kInstanceTag) >= hd.instance_count))
                return fail(c, PTR_E_STATE, "instances: root reference out of range");
            top_buf->top_view.nodes = (const float4*)((const uint8_t*)top_buf->ptr + hd.nodes_offset);
            top_buf->top_view.records = (const float4*)((const uint8_t*)top_buf->ptr + hd.records_offset);
            top_buf->top_view.root = hd.root;
            top_buf->top_view.count = hd.instance_count;
            top_buf->top_nodes = hd.node_count;
            top_buf->top_depth = hd.max_depth;
            top_buf->top_view_valid = true;
        }
        if (flat_buf->flat_triangles >= (1u << 28)) return fail(c, PTR_E_RANGE, "instances: the instanced layout holds 2^28 triangles or more");
        if (top_buf->top_depth < fv.max_depth + 1u) return fail(c, PTR_E_STATE, "instances: the stack bound in the header does not cover the instanced layout");
        fv.top = top_buf->top_view;
        fv.max_depth = top_buf->top_depth;
        sv.instance_records = fv.top.records;
        sv.instance_count = fv.top.count;
        sv.object_triangles = flat_buf->flat_triangles;
    }
    if ((rc = validate_scene(c, k.addr, handles, sv, flat_buf, true, no_alloc))) return rc;
    if ((rc = ensure_shade_records(c, k.addr, handles, &sv, no_alloc))) return rc;

    const uint32_t domain_w = gx * 8u < meta.width ? gx * 8u : meta.width;
    const uint32_t domain_h = gy * 8u < meta.height ? gy * 8u : meta.height;
    if (domain_w == 0 || domain_h == 0) return PTR_OK;

    // queue slots: whole 8x8 tiles of the owned rows (padded by one stripe); samples are indexed by pixel
    const uint32_t padded = ((meta.width + 7u) / 8u) * ((meta.height + 7u) / 8u + c->part_stripe / 8u) * 64u;
    const uint32_t paths = (padded > meta.width * meta.height ? padded : meta.width * meta.height) * sv.batch_samples;

    if (no_alloc) {
        // one frame, start to end on the context's stream, in this GPU's fixed scratch array
        if (paths > kStaticPaths)
            return fail(c, PTR_E_RANGE, "ptr_dispatch_static: a %u x %u frame needs %u queue slots, the fixed scratch holds %u", meta.width, meta.height, paths, kStaticPaths);
        if (fv.max_depth > kStaticMaxDepth)
            return fail(c, PTR_E_RANGE, "ptr_dispatch_static: the layout needs a traversal stack of %u entries, the fixed scratch covers %u", fv.max_depth, kStaticMaxDepth);
        ptr::WavefrontScratch w = c->static_scratch;
        w.alone_turns = (uint32_t)c->opt_extend_alone_turns;
        w.fault = c->fault_device;
        hipStream_t ts = c->stream;
        // the array keeps the last call's contents: the queue counters must start at zero
        PTR_HIP_CHECK(c, hipMemsetAsync(w.counters, 0, 64 * sizeof(uint32_t), ts));
        if (meta.max_bounce == 0) PTR_HIP_CHECK(c, hipMemsetAsync(w.samples, 0, (size_t)paths * 16, ts));
        PTR_HIP_CHECK(c, ptr::launch_wf_generate(ts, sv, w, domain_w, domain_h));
        for (uint32_t b = 0; b < meta.max_bounce; b++) {
            if (b + 1 >= ptr::kMaxBounceSlots) {
                const uint32_t s = (b + 1) % ptr::kMaxBounceSlots;
                PTR_HIP_CHECK(c, hipMemsetAsync(w.counters + s, 0, 4, ts));
                PTR_HIP_CHECK(c, hipMemsetAsync(w.counters + ptr::kMaxBounceSlots + s, 0, 4, ts));
            }
            PTR_HIP_CHECK(c, ptr::launch_wf_extend(ts, fv, w, b, sv.stats, c->opt_extend_refill, c->opt_extend_leaf_weight, c->opt_extend_leaf_all, c->opt_extend_inner_reps));
            PTR_HIP_CHECK(c, ptr::launch_wf_shade(ts, sv, w, b, meta.max_bounce, meta.min_bounce, domain_w, domain_h));
        }
        PTR_HIP_CHECK(c, ptr::launch_wf_accumulate(ts, sv, w, domain_w, domain_h));
        return PTR_OK;
    }

    // Which stream traces this frame: one of the auxiliary frame slots, or (counting / capturing runs, or the option
    // set to 1) the context's own stream after everything in flight has finished.
    const bool overlap = c->opt_frames_in_flight > 1 && !c->stats_on && !sv.capture;
    if (c->next_slot >= c->opt_frames_in_flight) c->next_slot = 0;
    ptr_ctx::FrameSlot& slot = c->slots[overlap ? c->next_slot : 0];
    if (!overlap && (rc = drain_frames(c))) return rc;
    if (overlap && (rc = ensure_slot(c, slot))) return rc;
    hipStream_t ts = overlap ? slot.stream : c->stream;

    if ((rc = ensure_scratch(c, slot, paths, fv.max_depth))) return rc;
    slot.scratch.alone_turns = (uint32_t)c->opt_extend_alone_turns;
    slot.scratch.fault = c->fault_device;
    const ptr::WavefrontScratch& w = slot.scratch;

    if (overlap) {
        // the slot's scratch is free once the accumulate step of its previous frame has run
        if (slot.in_flight) PTR_HIP_CHECK(c, hipStreamWaitEvent(ts, slot.consumed, 0));
        // Inputs are written on the context's stream.  The slot's stream waits for that stream only when one of this
        // frame's inputs changed since it last did: the wait also queues the frame behind the accumulate steps of the
        // frames before it, which is exactly the serialisation the slots exist to avoid.
        uint64_t newest = c->shared_epoch;
        for (int i = 1; i < 7; i++) if (c->buffers[handles[i]].generation > newest) newest = c->buffers[handles[i]].generation;
        if (flat_buf->generation > newest) newest = flat_buf->generation;
        if (top_buf && top_buf->generation > newest) newest = top_buf->generation;
        if (newest > slot.synced_epoch) {
            if (slot.ready_epoch < newest) {
                // This point of the context's stream is marked for EVERY slot that is behind, not only this one: once this
                // frame's accumulate step is on that stream, a mark made later would also stand behind it, and the next
                // frame -- on the other slot, and just as much behind after a ptr_sync or an upload -- would wait for this
                // whole frame instead of for the inputs.
                for (int i = 0; i < c->opt_frames_in_flight; i++) {
                    ptr_ctx::FrameSlot& o = c->slots[i];
                    if ((rc = ensure_slot(c, o))) return rc;
                    if (o.synced_epoch < c->write_epoch) {
                        PTR_HIP_CHECK(c, hipEventRecord(o.ready, c->stream));
                        o.ready_epoch = c->write_epoch;
                    }
                }
            }
            PTR_HIP_CHECK(c, hipStreamWaitEvent(ts, slot.ready, 0));
            slot.synced_epoch = slot.ready_epoch;
        }
    }
    PTR_HIP_CHECK(c, hipMemsetAsync(w.counters, 0, 64 * sizeof(uint32_t), ts));
    if (meta.max_bounce == 0) PTR_HIP_CHECK(c, hipMemsetAsync(w.samples, 0, (size_t)w.capacity * 16, ts));
    { ScopedProfile prof(c, "wavefront.generate", ts); PTR_HIP_CHECK(c, ptr::launch_wf_generate(ts, sv, w, domain_w, domain_h)); }
    for (uint32_t b = 0; b < meta.max_bounce; b++) {
        if (b + 1 >= ptr::kMaxBounceSlots) {  // recycle the counter slots of a bounce that finished long ago
            const uint32_t s = (b + 1) % ptr::kMaxBounceSlots;
            PTR_HIP_CHECK(c, hipMemsetAsync(w.counters + s, 0, 4, ts));
            PTR_HIP_CHECK(c, hipMemsetAsync(w.counters + ptr::kMaxBounceSlots + s, 0, 4, ts));
        }
        { ScopedProfile prof(c, "wavefront.extend", ts); PTR_HIP_CHECK(c, ptr::launch_wf_extend(ts, fv, w, b, sv.stats, c->opt_extend_refill, c->opt_extend_leaf_weight, c->opt_extend_leaf_all, c->opt_extend_inner_reps)); }
        { ScopedProfile prof(c, "wavefront.shade", ts); PTR_HIP_CHECK(c, ptr::launch_wf_shade(ts, sv, w, b, meta.max_bounce, meta.min_bounce, domain_w, domain_h)); }
    }
    if (overlap) {
        // the context's stream has already been told to wait for the previous frame's accumulate step, so this also
        // puts the accumulate steps of consecutive frames in dispatch order
        PTR_HIP_CHECK(c, hipEventRecord(slot.images, c->stream));
        PTR_HIP_CHECK(c, hipStreamWaitEvent(ts, slot.images, 0));
    }
    { ScopedProfile prof(c, "wavefront.accumulate", ts); PTR_HIP_CHECK(c, ptr::launch_wf_accumulate(ts, sv, w, domain_w, domain_h)); }
    if (overlap) {
        PTR_HIP_CHECK(c, hipEventRecord(slot.consumed, ts));
        PTR_HIP_CHECK(c, hipStreamWaitEvent(c->stream, slot.consumed, 0));
        slot.in_flight = true;
        for (int i = 0; i < 7; i++) slot.reads[i] = i == 0 ? 0 : handles[i];  // the metadata travels by value
        slot.reads[7] = bvh_buf->flat;
        slot.reads[8] = flat_buf->instances;
        c->next_slot = (c->next_slot + 1) % c->opt_frames_in_flight;
    }
    return PTR_OK;
}

// Frame scratch a chain of held-back dispatches may ask for ("dispatch.coalesce_scratch_mb", 8 GiB unless set): the queues, hits
// and samples of one slot grow with the samples of the dispatch that renders the chain, about 132 bytes per pixel and sample
// (17 GB for 64 samples at 1080p would be a surprise).  A chain that would pass it is rendered and a new one started.
constexpr uint64_t kScratchBytesPerPath = 132;

bool chain_fits(const ptr_ctx* c, const ptr::SceneMeta& meta, uint32_t count) {
    const uint64_t paths = (uint64_t)meta.width * meta.height * count;
    return paths < (1ull << 31) && paths * kScratchBytesPerPath <= ((uint64_t)c->opt_coalesce_scratch_mb << 20);
}

int flush_pending(ptr_ctx* c) {
    if (!c->pending.active) return PTR_OK;
    const ptr_ctx::Pending p = c->pending;
    c->pending.active = false;
    int rc = dispatch_compute(c, p.last, p.gx, p.gy, p.force, false, &p);
    if (rc == PTR_OK || p.count < 2u) return rc;
    // The batched dispatch was refused (its scratch could not be allocated, a precondition of batches that the single
    // dispatches do not have): the frames the caller issued are still owed.  They run one by one, each with the constants
    // and the metadata it was issued with -- dispatch i of the chain has the newest one's images, swapped when it lies an
    // odd number of dispatches before it.
    for (uint32_t i = 0; i < p.count; i++) {
        ptr_ctx::Pending one = p;
        one.count = 1;
        one.meta.sample_index = p.meta.sample_index + i;
        if ((p.count - 1u - i) & 1u) {
            memcpy(one.last + 56, p.last + 60, 4);
            memcpy(one.last + 60, p.last + 56, 4);
        }
        if ((rc = dispatch_compute(c, one.last, one.gx, one.gy, one.force, false, &one))) return rc;
    }
    return PTR_OK;
}

// A "compute" dispatch under "dispatch.coalesce" > 1: held back when it starts or continues an accumulation chain the batched
// dispatch can render (wavefront pipeline, one sample, no counting or capture run), otherwise run at once -- after the ones
// held back.  A dispatch that is held back reports nothing yet: what it would have reported comes from the call that runs it.
int dispatch_or_hold(ptr_ctx* c, const uint8_t constants[64], uint32_t gx, uint32_t gy, int force) {
    int rc = PTR_OK;
    const ParsedConstants k = parse_constants(constants);
    ptr::SceneMeta meta;
    ptr_handle handles[7] = {};
    bool can = c->opt_coalesce > 1 && c->opt_dispatch_samples == 1 && !c->stats_on && !c->capture && force != 1 &&
               read_meta(c, k.addr[0], &meta) == PTR_OK && meta.debug_bvh != 1u && meta.sample_index != 0u;
    for (int i = 0; can && i < 7; i++) can = buffer_at(c, k.addr[i], 1, &handles[i]) != nullptr;
    if (can && !c->buffers[handles[1]].flat) can = false;   // no flattened layout bound: the megakernel's frame, or an error to report now
    // a dispatch that names no usable image fails now, not when it would have run
    if (can && (check_storage_image(c, k.output_index, "output") || check_storage_image(c, k.accumulation_index, "accumulation"))) can = false;
    // what the batched dispatch asks beyond a single one: both images of one extent, and room for at least two samples
    if (can && (c->images[k.output_index].w != c->images[k.accumulation_index].w || c->images[k.output_index].h != c->images[k.accumulation_index].h)) can = false;
    if (can && !chain_fits(c, meta, 2u)) can = false;
    if (!can) {
        if ((rc = flush_pending(c))) return rc;
        return dispatch_compute(c, constants, gx, gy, force);
    }
    ptr_ctx::Pending& p = c->pending;
    if (p.active) {
        const ParsedConstants last = parse_constants(p.last);
        ptr::SceneMeta expect = p.meta;
        expect.sample_index = p.meta.sample_index + p.count;
        const bool next = p.gx == gx && p.gy == gy && p.force == force && p.meta.sample_index <= 0xffffffffu - p.count && chain_fits(c, p.meta, p.count + 1u) &&
                          !memcmp(&meta, &expect, sizeof meta) && !memcmp(constants, p.last, 56) &&
                          k.output_index == last.accumulation_index && k.accumulation_index == last.output_index;
        if (next) {
            memcpy(p.last, constants, 64);
            p.count++;
        } else if ((rc = flush_pending(c))) {
            return rc;
        }
    }
    if (!p.active) {
        p.active = true;
        memcpy(p.last, constants, 64);
        p.meta = meta;
        p.count = 1; p.gx = gx; p.gy = gy; p.force = force;
        for (int i = 0; i < 7; i++) p.reads[i] = i == 0 ? 0 : handles[i];   // the metadata was taken by value
        p.reads[7] = c->buffers[handles[1]].flat;
        const BufferRec* flat = find_buffer(c, p.reads[7]);
        p.reads[8] = flat ? flat->instances : 0;
    }
    if (p.count >= (uint32_t)c->opt_coalesce) return flush_pending(c);
    return PTR_OK;
}

}  // namespace

extern "C" {

int ptr_ctx_create(int device, ptr_ctx** out) {
    if (!out) return fail(nullptr, PTR_E_INVALID, "ptr_ctx_create: out is null");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) return fail(nullptr, PTR_E_HIP, "no HIP device: %s", hipGetErrorString(e));
    if (device < 0 || device >= count) return fail(nullptr, PTR_E_INVALID, "device %d out of range (%d devices)", device, count);
    ptr_ctx* c = new ptr_ctx();
    c->device = device;
    auto bail = [&](const char* what, hipError_t err) { fail(nullptr, PTR_E_HIP, "%s: %s", what, hipGetErrorString(err)); delete c; return PTR_E_HIP; };
    if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return bail("hipGetDeviceProperties", e);
    c->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if ((e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    c->stream = c->own_stream;
    if ((e = hipMalloc((void**)&c->d_images, kMaxImages * sizeof(ptr::ImageDesc))) != hipSuccess) return bail("hipMalloc(image table)", e);
    if ((e = hipMemset(c->d_images, 0, kMaxImages * sizeof(ptr::ImageDesc))) != hipSuccess) return bail("hipMemset(image table)", e);
    if ((e = hipHostMalloc((void**)&c->stage, (size_t)ptr_ctx::kStageSlots * kShadowLimit, hipHostMallocDefault)) != hipSuccess) return bail("hipHostMalloc(staging ring)", e);
    for (int i = 0; i < ptr_ctx::kStageSlots; i++)
        if ((e = hipEventCreateWithFlags(&c->stage_done[i], hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate(staging ring)", e);
    {
        // this GPU's copy of the fixed scratch array, carved up like an allocated frame scratch
        uint32_t* p = nullptr;
        if ((e = static_scratch_address(device, &p)) != hipSuccess) return bail("hipGetSymbolAddress(static scratch)", e);
        auto take = [&](size_t n) { uint32_t* r = p; p += n; return r; };
        ptr::WavefrontScratch& s = c->static_scratch;
        s.hits = (float4*)take((size_t)kStaticPaths * 4);
        s.samples = (float4*)take((size_t)kStaticPaths * 4);
        for (int g = 0; g < 2; g++) {
            s.ray_o[g] = (float4*)take((size_t)kStaticPaths * 4);
            s.ray_d[g] = (float4*)take((size_t)kStaticPaths * 4);
            s.path_t[g] = (float4*)take((size_t)kStaticPaths * 4);
        }
        s.hit_instance = (int32_t*)take(kStaticPaths);
        s.counters = take(64);
        s.stack_spill = (int32_t*)take(kStaticSpillWords);
        s.capacity = kStaticPaths;
        s.extend_blocks = kStaticExtendBlocks;
        s.spill_entries = kStaticSpill;
    }
    if ((e = hipHostMalloc((void**)&c->fault, 64, hipHostMallocMapped)) != hipSuccess) return bail("hipHostMalloc(fault word)", e);
    *c->fault = 0u;
    if ((e = hipHostGetDevicePointer((void**)&c->fault_device, (void*)c->fault, 0)) != hipSuccess) return bail("hipHostGetDevicePointer(fault word)", e);
    if ((e = hipMalloc((void**)&c->d_stats, 8 * sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc(stats)", e);
    if ((e = hipMemset(c->d_stats, 0, 8 * sizeof(unsigned long long))) != hipSuccess) return bail("hipMemset(stats)", e);
    *out = c;
    return PTR_OK;
}

int ptr_ctx_destroy(ptr_ctx* c) {
    if (!c) return PTR_E_INVALID;
    (void)hipSetDevice(c->device);
    (void)flush_pending(c);   // dispatches still held back write their images like any other
    for (auto& slot : c->slots) if (slot.stream) (void)hipStreamSynchronize(slot.stream);
    (void)hipStreamSynchronize(c->stream);
    drain_profile(c);
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    for (auto& slot : c->slots) {
        if (slot.scratch_block) (void)hipFree(slot.scratch_block);
        for (hipEvent_t e : { slot.ready, slot.images, slot.consumed }) if (e) (void)hipEventDestroy(e);
        if (slot.stream) (void)hipStreamDestroy(slot.stream);
    }
    for (auto& b : c->buffers) if (b.alive && b.ptr) (void)hipFree(b.ptr);
    for (auto& im : c->images) if (im.alive && im.owned && im.ptr) (void)hipFree(im.ptr);
    for (auto& im : c->images) if (im.alive && im.footprints) (void)hipFree(im.footprints);
    if (c->shade_records) (void)hipFree(c->shade_records);
    if (c->unorm8_stage) (void)hipFree(c->unorm8_stage);
    if (c->d_validate) (void)hipFree(c->d_validate);
    if (c->d_images) (void)hipFree(c->d_images);
    if (c->d_stats) (void)hipFree(c->d_stats);
    for (auto e : c->stage_done) if (e) (void)hipEventDestroy(e);
    if (c->stage) (void)hipHostFree(c->stage);
    if (c->fault) (void)hipHostFree((void*)c->fault);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return PTR_OK;
}

int ptr_ctx_set_stream(ptr_ctx* c, void* hip_stream) {
    if (!c) return PTR_E_INVALID;
    int rc = enter(c); if (rc) return rc;
    if ((rc = drain_frames(c))) return rc;
    PTR_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    for (auto& slot : c->slots) slot.in_flight = false;  // their `consumed` events belong to the old stream, which is idle now
    c->shared_epoch = ++c->write_epoch;
    // A process has few hardware queues (4 here) and streams that share one run strictly one after the other: the
    // context's own stream is given up while the caller's stream is in use, so that the frame streams find a free queue.
    if (hip_stream && c->own_stream) {
        PTR_HIP_CHECK(c, hipStreamDestroy(c->own_stream));
        c->own_stream = nullptr;
    } else if (!hip_stream && !c->own_stream) {
        PTR_HIP_CHECK(c, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    }
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return PTR_OK;
}

const char* ptr_last_error(const ptr_ctx* c) { return c ? c->error.c_str() : g_create_error.c_str(); }

int ptr_flush(ptr_ctx* c) {
    if (!c) return PTR_E_INVALID;
    return enter(c);
}

int ptr_sync(ptr_ctx* c) {
    if (!c) return PTR_E_INVALID;
    int rc = enter(c); if (rc) return rc;
    if ((rc = drain_frames(c))) return rc;
    PTR_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    // whatever the caller wrote into device memory by other means before this point is ordered before later frames
    c->shared_epoch = ++c->write_epoch;
    return take_fault(c);
}

int ptr_buffer_create(ptr_ctx* c, size_t size, ptr_handle* out) {
    if (!c || !out) return PTR_E_INVALID;
    if (size == 0) return fail(c, PTR_E_INVALID, "ptr_buffer_create: size 0");
    int rc = enter(c); if (rc) return rc;
    BufferRec b;
    PTR_HIP_CHECK(c, hipMalloc(&b.ptr, size));
    b.size = size; b.alive = true;
    b.generation = ++c->write_epoch;
    if (size <= kShadowLimit) b.shadow.assign(size, 0);
    if (size <= kShadowLimit) PTR_HIP_CHECK(c, hipMemsetAsync(b.ptr, 0, size, c->stream));
    c->buffers.push_back(std::move(b));
    *out = (ptr_handle)(c->buffers.size() - 1);
    return PTR_OK;
}

int ptr_buffer_destroy(ptr_ctx* c, ptr_handle h) {
    if (!c) return PTR_E_INVALID;
    BufferRec* b = find_buffer(c, h);
    if (!b) return fail(c, PTR_E_INVALID, "ptr_buffer_destroy: unknown buffer %u", h);
    int rc = enter(c); if (rc) return rc;
    if ((rc = drain_frames(c))) return rc;
    PTR_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    if (b->flat_is_auto && b->flat) {
        // the layout "bvh.auto_flatten" built for this node buffer lives behind a handle the caller never saw: it goes with the buffer
        BufferRec& built = c->buffers[b->flat];
        PTR_HIP_CHECK(c, hipFree(built.ptr));
        built = BufferRec();
    }
    PTR_HIP_CHECK(c, hipFree(b->ptr));
    *b = BufferRec();
    for (auto& o : c->buffers) { if (o.flat == h) o.flat = 0; if (o.instances == h) o.instances = 0; }
    if (c->capture == h) c->capture = 0;
    return PTR_OK;
}

int ptr_buffer_write(ptr_ctx* c, ptr_handle h, const void* data, size_t offset, size_t size) {
    if (!c) return PTR_E_INVALID;
    BufferRec* b = find_buffer(c, h);
    if (!b) return fail(c, PTR_E_INVALID, "ptr_buffer_write: unknown buffer %u", h);
    if (!data && size) return fail(c, PTR_E_INVALID, "ptr_buffer_write: null data");
    if (offset > b->size || size > b->size - offset) return fail(c, PTR_E_RANGE, "ptr_buffer_write: [%zu, +%zu) outside a %zu-byte buffer", offset, size, b->size);
    if (size == 0) return PTR_OK;
    int rc = bind_device(c); if (rc) return rc;
    // dispatches held back read their metadata when they were issued: the per-frame metadata write does not disturb them
    if (c->pending.active)
        for (ptr_handle r : c->pending.reads)
            if (r == h) { if ((rc = flush_pending(c))) return rc; break; }
    if (frame_reads(c, h) && (rc = drain_frames(c))) return rc;  // a frame in flight still reads this buffer
    if (!b->shadow.empty()) memcpy(b->shadow.data() + offset, data, size);
    b->flat_view_valid = false;
    b->top_view_valid = false;
    b->generation = ++c->write_epoch;
    if (size <= kShadowLimit && c->stage) {
        const int slot = c->stage_next;
        c->stage_next = (slot + 1) % ptr_ctx::kStageSlots;
        PTR_HIP_CHECK(c, hipEventSynchronize(c->stage_done[slot]));  // the copy that last used this slot has finished
        uint8_t* pinned = c->stage + (size_t)slot * kShadowLimit;
        memcpy(pinned, data, size);
        PTR_HIP_CHECK(c, hipMemcpyAsync((uint8_t*)b->ptr + offset, pinned, size, hipMemcpyHostToDevice, c->stream));
        PTR_HIP_CHECK(c, hipEventRecord(c->stage_done[slot], c->stream));
        return PTR_OK;
    }
    // Buffer::write is a memcpy into mapped memory in the reference: the caller may reuse `data` at once.
    PTR_HIP_CHECK(c, hipMemcpyAsync((uint8_t*)b->ptr + offset, data, size, hipMemcpyHostToDevice, c->stream));
    PTR_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return PTR_OK;
}

int ptr_buffer_read(ptr_ctx* c, ptr_handle h, void* dst, size_t offset, size_t size) {
    if (!c) return PTR_E_INVALID;
    BufferRec* b = find_buffer(c, h);
    if (!b) return fail(c, PTR_E_INVALID, "ptr_buffer_read: unknown buffer %u", h);
    if (!dst && size) return fail(c, PTR_E_INVALID, "ptr_buffer_read: null dst");
    if (offset > b->size || size > b->size - offset) return fail(c, PTR_E_RANGE, "ptr_buffer_read: [%zu, +%zu) outside a %zu-byte buffer", offset, size, b->size);
    if (size == 0) return PTR_OK;
    int rc = enter(c); if (rc) return rc;
    PTR_HIP_CHECK(c, hipMemcpyAsync(dst, (const uint8_t*)b->ptr + offset, size, hipMemcpyDeviceToHost, c->stream));
    PTR_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return PTR_OK;
}

int ptr_buffer_address(ptr_ctx* c, ptr_handle h, uint64_t* addr) {
    if (!c || !addr) return PTR_E_INVALID;
    BufferRec* b = find_buffer(c, h);
    if (!b) return fail(c, PTR_E_INVALID, "ptr_buffer_address: unknown buffer %u", h);
    *addr = (uint64_t)(uintptr_t)b->ptr;
    return PTR_OK;
}

int ptr_sampler_create(ptr_ctx* c, int filter, int address_mode, ptr_handle* out) {
    if (!c || !out) return PTR_E_INVALID;
    if (filter != PTR_FILTER_LINEAR || address_mode != PTR_ADDRESS_REPEAT)
        return fail(c, PTR_E_UNSUPPORTED, "ptr_sampler_create: only LINEAR filtering with REPEAT addressing is implemented");
    SamplerRec s; s.filter = filter; s.address = address_mode; s.alive = true;
    c->samplers.push_back(s);
    *out = (ptr_handle)(c->samplers.size() - 1);
    return PTR_OK;
}

static int new_image(ptr_ctx* c, void* memory, bool owned, uint32_t w, uint32_t h, int format, ptr_handle sampler, ptr_handle* out) {
    if (c->images.size() >= kMaxImages) { if (owned) (void)hipFree(memory); return fail(c, PTR_E_RANGE, "bindless image table is full (%u)", kMaxImages); }
    ImageRec r; r.ptr = memory; r.w = w; r.h = h; r.format = format; r.sampler = sampler; r.owned = owned; r.alive = true;
    c->images.push_back(r);
    *out = (ptr_handle)(c->images.size() - 1);
    return upload_image_desc(c, *out);
}

int ptr_image_create(ptr_ctx* c, uint32_t w, uint32_t h, int format, ptr_handle sampler, ptr_handle* out) {
    if (!c || !out) return PTR_E_INVALID;
    const size_t ts = texel_size(format);
    if (ts == 0) return fail(c, PTR_E_UNSUPPORTED, "ptr_image_create: format %d (RGBA8_UNORM and RGBA32_SFLOAT are implemented)", format);
    if (w == 0 || h == 0) return fail(c, PTR_E_INVALID, "ptr_image_create: zero extent");
    if (sampler != 0 && (sampler >= c->samplers.size() || !c->samplers[sampler].alive)) return fail(c, PTR_E_INVALID, "ptr_image_create: unknown sampler %u", sampler);
    int rc = enter(c); if (rc) return rc;
    void* mem = nullptr;
    PTR_HIP_CHECK(c, hipMalloc(&mem, (size_t)w * h * ts));
    PTR_HIP_CHECK(c, hipMemsetAsync(mem, 0, (size_t)w * h * ts, c->stream));
    void* footprints = nullptr;
    if (format == PTR_FORMAT_RGBA8_UNORM) {   // textures are sampled from a copy laid out per lookup (4 x the texels), kept in step by ptr_image_upload
        const size_t footprint_bytes = (size_t)((w + 1u) / 2u) * ((h + 1u) / 2u) * 64u;
        if (hipMalloc(&footprints, footprint_bytes) != hipSuccess) { (void)hipFree(mem); return fail(c, PTR_E_HIP, "ptr_image_create: out of device memory for the lookup copy"); }
        PTR_HIP_CHECK(c, hipMemsetAsync(footprints, 0, footprint_bytes, c->stream));
    }
    const int rc_new = new_image(c, mem, true, w, h, format, sampler, out);
    if (rc_new != PTR_OK) { if (footprints) (void)hipFree(footprints); return rc_new; }
    if (footprints) { c->images[*out].footprints = footprints; return upload_image_desc(c, *out); }
    return PTR_OK;
}

int ptr_image_wrap(ptr_ctx* c, void* device_memory, uint32_t w, uint32_t h, int format, ptr_handle* out) {
    if (!c || !out) return PTR_E_INVALID;
    if (texel_size(format) == 0) return fail(c, PTR_E_UNSUPPORTED, "ptr_image_wrap: format %d", format);
    if (!device_memory || ((uintptr_t)device_memory & 15u)) return fail(c, PTR_E_INVALID, "ptr_image_wrap: memory must be non-null and 16-byte aligned");
    if (w == 0 || h == 0) return fail(c, PTR_E_INVALID, "ptr_image_wrap: zero extent");
    int rc = enter(c); if (rc) return rc;
    return new_image(c, device_memory, false, w, h, format, 0, out);
}

int ptr_image_destroy(ptr_ctx* c, ptr_handle h) {
    if (!c) return PTR_E_INVALID;
    ImageRec* im = find_image(c, h);
    if (!im) return fail(c, PTR_E_INVALID, "ptr_image_destroy: unknown image %u", h);
    int rc = enter(c); if (rc) return rc;
    if ((rc = drain_frames(c))) return rc;
    PTR_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    if (im->owned) PTR_HIP_CHECK(c, hipFree(im->ptr));
    if (im->footprints) PTR_HIP_CHECK(c, hipFree(im->footprints));
    *im = ImageRec();
    return upload_image_desc(c, h);
}

int ptr_image_upload(ptr_ctx* c, ptr_handle h, const void* texels) {
    if (!c) return PTR_E_INVALID;
    ImageRec* im = find_image(c, h);
    if (!im) return fail(c, PTR_E_INVALID, "ptr_image_upload: unknown image %u", h);
    if (!texels) return fail(c, PTR_E_INVALID, "ptr_image_upload: null texels");
    int rc = enter(c); if (rc) return rc;
    if ((rc = drain_frames(c))) return rc;  // frames in flight may sample this image
    PTR_HIP_CHECK(c, hipMemcpyAsync(im->ptr, texels, (size_t)im->w * im->h * texel_size(im->format), hipMemcpyHostToDevice, c->stream));
    if (im->footprints) PTR_HIP_CHECK(c, ptr::launch_footprints_rgba8(c->stream, im->ptr, im->footprints, im->w, im->h));
    PTR_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    c->shared_epoch = ++c->write_epoch;
    return PTR_OK;
}

int ptr_image_read(ptr_ctx* c, ptr_handle h, void* texels) {
    if (!c) return PTR_E_INVALID;
    ImageRec* im = find_image(c, h);
    if (!im) return fail(c, PTR_E_INVALID, "ptr_image_read: unknown image %u", h);
    if (!texels) return fail(c, PTR_E_INVALID, "ptr_image_read: null texels");
    int rc = enter(c); if (rc) return rc;
    PTR_HIP_CHECK(c, hipMemcpyAsync(texels, im->ptr, (size_t)im->w * im->h * texel_size(im->format), hipMemcpyDeviceToHost, c->stream));
    PTR_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return take_fault(c);
}

int ptr_image_read_unorm8(ptr_ctx* c, ptr_handle h, void* rgba8) {
    if (!c) return PTR_E_INVALID;
    ImageRec* im = find_image(c, h);
    if (!im) return fail(c, PTR_E_INVALID, "ptr_image_read_unorm8: unknown image %u", h);
    if (!rgba8) return fail(c, PTR_E_INVALID, "ptr_image_read_unorm8: null destination");
    if (im->format != PTR_FORMAT_RGBA32_SFLOAT) return fail(c, PTR_E_UNSUPPORTED, "ptr_image_read_unorm8: the image must be RGBA32_SFLOAT");
    int rc = enter(c); if (rc) return rc;
    const size_t bytes = (size_t)im->w * im->h * 4;
    if (bytes == 0) return PTR_OK;
    if (bytes > c->unorm8_stage_bytes) {
        PTR_HIP_CHECK(c, hipStreamSynchronize(c->stream));
        if (c->unorm8_stage) PTR_HIP_CHECK(c, hipFree(c->unorm8_stage));
        c->unorm8_stage = nullptr; c->unorm8_stage_bytes = 0;
        PTR_HIP_CHECK(c, hipMalloc(&c->unorm8_stage, bytes));
        c->unorm8_stage_bytes = bytes;
    }
    PTR_HIP_CHECK(c, ptr::launch_unorm8_pack(c->stream, im->ptr, im->w * im->h, c->unorm8_stage));
    PTR_HIP_CHECK(c, hipMemcpyAsync(rgba8, c->unorm8_stage, bytes, hipMemcpyDeviceToHost, c->stream));
    PTR_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return take_fault(c);
}

int ptr_device_function(ptr_ctx* c, const char* name, const uint32_t* in, uint32_t in_words, uint32_t* out, uint32_t out_words,
                        uint32_t count) {
    if (!c) return PTR_E_INVALID;
    if (!name || !in || !out) return fail(c, PTR_E_INVALID, "ptr_device_function: null argument");
    uint32_t want_in = 0, want_out = 0;
    const int fn = ptr::unit_function_lookup(name, &want_in, &want_out);
    if (fn < 0) return fail(c, PTR_E_UNSUPPORTED, "ptr_device_function: unknown function '%s'", name);
    if (in_words != want_in || out_words != want_out)
        return fail(c, PTR_E_INVALID, "ptr_device_function: '%s' takes %u words and returns %u, got %u and %u", name, want_in, want_out, in_words, out_words);
    if (count == 0) return PTR_OK;
    if (count > (1u << 24)) return fail(c, PTR_E_RANGE, "ptr_device_function: at most %u records per call", 1u << 24);
    int rc = enter(c); if (rc) return rc;
    uint32_t *d_in = nullptr, *d_out = nullptr;
    const size_t in_bytes = (size_t)count * in_words * 4, out_bytes = (size_t)count * out_words * 4;
    PTR_HIP_CHECK(c, hipMalloc((void**)&d_in, in_bytes));
    hipError_t e = hipMalloc((void**)&d_out, out_bytes);
    if (e == hipSuccess) e = hipMemcpyAsync(d_in, in, in_bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_out, 0, out_bytes, c->stream);
    if (e == hipSuccess) e = ptr::launch_unit_function(c->stream, fn, d_in, in_words, d_out, out_words, count);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) return fail(c, PTR_E_HIP, "ptr_device_function: %s", hipGetErrorString(e));
    return PTR_OK;
}

int ptr_image_index(ptr_ctx* c, ptr_handle h, uint32_t* index) {
    if (!c || !index) return PTR_E_INVALID;
    if (!find_image(c, h)) return fail(c, PTR_E_INVALID, "ptr_image_index: unknown image %u", h);
    *index = h;
    return PTR_OK;
}

int ptr_sampler_index(ptr_ctx* c, ptr_handle h, uint32_t* index) {
    if (!c || !index) return PTR_E_INVALID;
    if (h == 0 || h >= c->samplers.size() || !c->samplers[h].alive) return fail(c, PTR_E_INVALID, "ptr_sampler_index: unknown sampler %u", h);
    *index = h;
    return PTR_OK;
}

int ptr_image_address(ptr_ctx* c, ptr_handle h, uint64_t* addr) {
    if (!c || !addr) return PTR_E_INVALID;
    ImageRec* im = find_image(c, h);
    if (!im) return fail(c, PTR_E_INVALID, "ptr_image_address: unknown image %u", h);
    *addr = (uint64_t)(uintptr_t)im->ptr;
    return PTR_OK;
}

int ptr_bvh_bind_flat(ptr_ctx* c, ptr_handle threaded_nodes, ptr_handle flat) {
    if (!c) return PTR_E_INVALID;
    BufferRec* b = find_buffer(c, threaded_nodes);
    if (!b) return fail(c, PTR_E_INVALID, "ptr_bvh_bind_flat: unknown node buffer %u", threaded_nodes);
    if (flat != 0) {
        BufferRec* f = find_buffer(c, flat);
        if (!f) return fail(c, PTR_E_INVALID, "ptr_bvh_bind_flat: unknown flat buffer %u", flat);
        if (f->size < 64) return fail(c, PTR_E_INVALID, "ptr_bvh_bind_flat: flat buffer shorter than the 64-byte header");
    }
    int rc = enter(c); if (rc) return rc;
    if ((rc = drain_frames(c))) return rc;
    if (b->flat_is_auto && b->flat && b->flat != flat) {   // the caller's layout replaces the one the library built
        PTR_HIP_CHECK(c, hipStreamSynchronize(c->stream));
        PTR_HIP_CHECK(c, hipFree(c->buffers[b->flat].ptr));
        c->buffers[b->flat] = BufferRec();
    }
    b->flat = flat;
    b->flat_is_auto = false;
    return PTR_OK;
}

int ptr_bvh_bind_instances(ptr_ctx* c, ptr_handle flat, ptr_handle instances) {
    if (!c) return PTR_E_INVALID;
    BufferRec* f = find_buffer(c, flat);
    if (!f) return fail(c, PTR_E_INVALID, "ptr_bvh_bind_instances: unknown flat buffer %u", flat);
    if (instances != 0) {
        BufferRec* t = find_buffer(c, instances);
        if (!t) return fail(c, PTR_E_INVALID, "ptr_bvh_bind_instances: unknown instance buffer %u", instances);
        if (t->size < 64) return fail(c, PTR_E_INVALID, "ptr_bvh_bind_instances: instance buffer shorter than the 64-byte header");
        if (instances == flat) return fail(c, PTR_E_INVALID, "ptr_bvh_bind_instances: the instance buffer cannot be the flat buffer itself");
    }
    int rc = enter(c); if (rc) return rc;
    if ((rc = drain_frames(c))) return rc;
    f->instances = instances;
    return PTR_OK;
}

int ptr_bvh_build_flat(ptr_ctx* c, ptr_handle indices, ptr_handle positions, uint32_t triangle_count, ptr_handle* flat_out) {
    if (!c) return PTR_E_INVALID;
    if (!flat_out) return fail(c, PTR_E_INVALID, "ptr_bvh_build_flat: flat_out is null");
    BufferRec* ib = find_buffer(c, indices);
    BufferRec* pb = find_buffer(c, positions);
    if (!ib || !pb) return fail(c, PTR_E_INVALID, "ptr_bvh_build_flat: unknown index or position buffer");
    if (triangle_count == 0) return fail(c, PTR_E_INVALID, "ptr_bvh_build_flat: no triangles");
    if (triangle_count >= (1u << 29)) return fail(c, PTR_E_RANGE, "ptr_bvh_build_flat: too many triangles for the leaf reference encoding");
    const size_t index_bytes = (size_t)triangle_count * (size_t)c->opt_index_words * 4;
    if (ib->size < index_bytes) return fail(c, PTR_E_RANGE, "ptr_bvh_build_flat: the index buffer holds fewer than %u triangles of %d words", triangle_count, c->opt_index_words);
    const uint64_t vertices = pb->size / 12u;
    if (vertices == 0) return fail(c, PTR_E_RANGE, "ptr_bvh_build_flat: the position buffer holds no vertex");
    int rc = enter(c); if (rc) return rc;
    ptr::SceneView sv{};
    sv.indices = (const uint32_t*)ib->ptr;
    sv.positions = (const float*)pb->ptr;
    sv.index_words = (uint32_t)c->opt_index_words;
    const size_t bytes = ptr::flat_build_blob_bytes(triangle_count);
    BufferRec b;
    PTR_HIP_CHECK(c, hipMalloc(&b.ptr, bytes));
    int status = 0;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (c->profile_on) { t0 = take_event(c); t1 = take_event(c); if (t0) (void)hipEventRecord(t0, c->stream); }
    const uint32_t vertex_count = (uint32_t)(vertices > 0xffffffffull ? 0xffffffffull : vertices);
    hipError_t e = ptr::build_flat_bvh(c->stream, sv, triangle_count, vertex_count, b.ptr, bytes, (float)c->opt_build_traversal_cost * 0.01f, c->opt_build_cluster_radius, &status);
    // A clustered tree can fail where the radix tree does not: clustering that does not finish within its round limit (-6: boxes
    // that each prefer the next one along a long chain), or a tree as deep as its round count -- boxes that grow geometrically
    // along the Morton curve merge one pair per round -- that exceeds the traversal stack or the level limit (-5, -2).  The
    // radix tree's depth is bounded by the key bits plus log n: every failed clustered build is tried again with it.
    if (e == hipSuccess && status != 0 && c->opt_build_cluster_radius != 0)
        e = ptr::build_flat_bvh(c->stream, sv, triangle_count, vertex_count, b.ptr, bytes, (float)c->opt_build_traversal_cost * 0.01f, 0, &status);
    if (c->profile_on && t0 && t1) { (void)hipEventRecord(t1, c->stream); c->profile["bvh.build_flat"].pending.emplace_back(t0, t1); }
    if (e != hipSuccess || status != 0) {
        (void)hipFree(b.ptr);
        if (e != hipSuccess) return fail(c, PTR_E_HIP, "ptr_bvh_build_flat: %s", hipGetErrorString(e));
        return fail(c, status == -5 ? PTR_E_RANGE : PTR_E_STATE, status == -5 ? "ptr_bvh_build_flat: the tree is deeper than the traversal stack (use the host builder)"
                                                                                : "ptr_bvh_build_flat: the tree could not be built (status %d)", status);
    }
    b.size = bytes; b.alive = true;
    b.generation = ++c->write_epoch;
    c->buffers.push_back(std::move(b));
    *flat_out = (ptr_handle)(c->buffers.size() - 1);
    return PTR_OK;
}

int ptr_bvh_flat_set_reference_ranks(ptr_ctx* c, ptr_handle flat, ptr_handle threaded_nodes, uint32_t triangle_count) {
    if (!c) return PTR_E_INVALID;
    BufferRec* f = find_buffer(c, flat);
    BufferRec* t = find_buffer(c, threaded_nodes);
    if (!f || !t) return fail(c, PTR_E_INVALID, "ptr_bvh_flat_set_reference_ranks: unknown flat or node buffer");
    if (triangle_count == 0) return fail(c, PTR_E_INVALID, "ptr_bvh_flat_set_reference_ranks: no triangles");
    int rc = enter(c); if (rc) return rc;
    if ((rc = drain_frames(c))) return rc;
    struct Header { uint32_t magic, version, node_count, triangle_count; uint64_t nodes_offset, triangles_offset; int32_t root; uint32_t max_depth; float bounds[6]; } hd;
    if (f->size < sizeof hd) return fail(c, PTR_E_STATE, "ptr_bvh_flat_set_reference_ranks: blob shorter than its header");
    PTR_HIP_CHECK(c, hipMemcpyAsync(&hd, f->ptr, sizeof hd, hipMemcpyDeviceToHost, c->stream));
    PTR_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    if (hd.magic != 0x46525450u || hd.version != 2u || (hd.triangles_offset & 63u) || hd.triangles_offset + (uint64_t)hd.triangle_count * 64 > f->size)
        return fail(c, PTR_E_STATE, "ptr_bvh_flat_set_reference_ranks: not a flattened layout");
    // the reference's builder emits 2N - 1 nodes for N triangles (src/bvh.cpp:48-86); a larger buffer's tail is not part of the tree
    uint64_t nodes = t->size / sizeof(ptr::ThreadedNode);
    if (nodes > 2ull * triangle_count - 1ull) nodes = 2ull * triangle_count - 1ull;
    PTR_HIP_CHECK(c, ptr::apply_reference_ranks(c->stream, (const ptr::ThreadedNode*)t->ptr, (uint32_t)nodes,
                                                (float4*)((uint8_t*)f->ptr + hd.triangles_offset), hd.triangle_count, triangle_count));
    f->generation = ++c->write_epoch;
    return PTR_OK;
}

int ptr_dispatch(ptr_ctx* c, const char* kernel, const uint8_t constants[64], uint32_t gx, uint32_t gy, uint32_t gz) {
    if (!c) return PTR_E_INVALID;
    if (!kernel || !constants) return fail(c, PTR_E_INVALID, "ptr_dispatch: null kernel name or constants");
    int rc = bind_device(c); if (rc) return rc;
    if (gx == 0 || gy == 0 || gz == 0) return PTR_OK;
    if ((uint64_t)gx * gy > 0x7fffffffull) return fail(c, PTR_E_RANGE, "ptr_dispatch: too many workgroups");
    // the shaders only read gl_GlobalInvocationID.xy: z slices would repeat the same work, one slice is run
    if (!strcmp(kernel, "compute")) return dispatch_or_hold(c, constants, gx, gy, 0);
    if (!strcmp(kernel, "compute.megakernel")) return dispatch_or_hold(c, constants, gx, gy, 1);
    if (!strcmp(kernel, "compute.wavefront")) return dispatch_or_hold(c, constants, gx, gy, 2);
    if ((rc = flush_pending(c))) return rc;
    if (!strcmp(kernel, "tonemapping")) {
        const ParsedConstants k = parse_constants(constants);
        if (!buffer_at(c, k.addr[0], sizeof(ptr::SceneMeta))) return fail(c, PTR_E_STATE, "constant block: scene metadata address is not inside a registered buffer");
        if ((rc = check_storage_image(c, k.output_index, "output"))) return rc;
        if ((rc = check_storage_image(c, k.accumulation_index, "accumulation"))) return rc;
        ScopedProfile prof(c, "tonemapping");
        PTR_HIP_CHECK(c, ptr::launch_tonemap(c->stream, (const ptr::SceneMeta*)(uintptr_t)k.addr[0], c->d_images, (uint32_t)c->images.size(),
                                             k.output_index, k.accumulation_index, gx, gy));
        return PTR_OK;
    }
    if (!strcmp(kernel, "linearize") || !strcmp(kernel, "resolve")) {
        const ParsedConstants k = parse_constants(constants);
        if ((rc = check_storage_image(c, k.output_index, "output"))) return rc;
        if ((rc = check_storage_image(c, k.accumulation_index, "accumulation"))) return rc;
        const ImageRec& a = c->images[k.accumulation_index];
        const ImageRec& o = c->images[k.output_index];
        if (a.w != o.w || a.h != o.h) return fail(c, PTR_E_STATE, "%s: output and accumulation images differ in size", kernel);
        ScopedProfile prof(c, kernel[0] == 'l' ? "linearize" : "resolve");
        PTR_HIP_CHECK(c, ptr::launch_linearize_resolve(c->stream, c->d_images, k.output_index, k.accumulation_index, a.w * a.h, kernel[0] == 'r'));
        return PTR_OK;
    }
    return fail(c, PTR_E_INVALID, "ptr_dispatch: unknown kernel '%s' (compute, compute.megakernel, compute.wavefront, tonemapping, linearize, resolve)", kernel);
}

int ptr_dispatch_static(ptr_ctx* c, const char* kernel, const uint8_t constants[64], uint32_t gx, uint32_t gy, uint32_t gz) {
    if (!c) return PTR_E_INVALID;
    if (!kernel || !constants) return fail(c, PTR_E_INVALID, "ptr_dispatch_static: null kernel name or constants");
    int rc = bind_device(c); if (rc) return rc;
    // running held-back dispatches here would allocate their scratch: the caller runs them first, where it may
    if (c->pending.active) return fail(c, PTR_E_STATE, "ptr_dispatch_static: dispatches are being held back (\"dispatch.coalesce\"); call ptr_flush or ptr_sync first");
    if (gx == 0 || gy == 0 || gz == 0) return PTR_OK;
    if ((uint64_t)gx * gy > 0x7fffffffull) return fail(c, PTR_E_RANGE, "ptr_dispatch_static: too many workgroups");
    if (!strcmp(kernel, "compute")) return dispatch_compute(c, constants, gx, gy, 0, true);
    if (!strcmp(kernel, "compute.megakernel")) return dispatch_compute(c, constants, gx, gy, 1, true);
    if (!strcmp(kernel, "compute.wavefront")) return dispatch_compute(c, constants, gx, gy, 2, true);
    return fail(c, PTR_E_INVALID, "ptr_dispatch_static: unknown kernel '%s' (compute, compute.megakernel, compute.wavefront)", kernel);
}

int ptr_set_partition(ptr_ctx* c, uint32_t rank, uint32_t world, uint32_t stripe_rows) {
    if (!c) return PTR_E_INVALID;
    if (world == 0 || rank >= world) return fail(c, PTR_E_INVALID, "ptr_set_partition: rank %u / world %u", rank, world);
    if (stripe_rows == 0 || (stripe_rows & 7u) || stripe_rows > 65536u) return fail(c, PTR_E_INVALID, "ptr_set_partition: stripe_rows must be a positive multiple of 8, 65536 at most");
    int rc = enter(c); if (rc) return rc;
    c->part_rank = rank; c->part_world = world; c->part_stripe = stripe_rows;
    return PTR_OK;
}

static int stripes_copy(ptr_ctx* c, const char* who, ptr_handle image, uint32_t rank, uint32_t ranks, uint32_t world, uint32_t stripe_rows, ptr_handle buffer, bool unpack) {
    if (!c) return PTR_E_INVALID;
    ImageRec* im = find_image(c, image);
    BufferRec* b = find_buffer(c, buffer);
    if (!im || !b) return fail(c, PTR_E_INVALID, "%s: unknown image or buffer", who);
    if (im->format != PTR_FORMAT_RGBA32_SFLOAT) return fail(c, PTR_E_UNSUPPORTED, "%s: the image must be RGBA32_SFLOAT", who);
    if (world == 0 || rank >= world || stripe_rows == 0 || (stripe_rows & 7u) || stripe_rows > 65536u) return fail(c, PTR_E_INVALID, "%s: rank %u / world %u / stripe_rows %u", who, rank, world, stripe_rows);
    const uint64_t stripes = ((uint64_t)im->h + stripe_rows - 1u) / stripe_rows;
    const uint64_t chunk_bytes = ((stripes + world - 1u) / world) * stripe_rows * im->w * 16u;
    if (chunk_bytes * ranks > b->size) return fail(c, PTR_E_RANGE, "%s: the buffer holds %zu bytes, %llu needed", who, b->size, (unsigned long long)(chunk_bytes * ranks));
    if (chunk_bytes * ranks / 16u > 0xffffffffull * 256ull) return fail(c, PTR_E_RANGE, "%s: frame too large", who);
    int rc = enter(c); if (rc) return rc;
    PTR_HIP_CHECK(c, ptr::launch_stripes_copy(c->stream, (float4*)im->ptr, im->w, im->h, stripe_rows, world, rank, ranks, (float4*)b->ptr, unpack));
    if (!unpack) b->generation = ++c->write_epoch;
    return PTR_OK;
}

int ptr_image_pack_stripes(ptr_ctx* c, ptr_handle image, uint32_t rank, uint32_t world, uint32_t stripe_rows, ptr_handle packed) {
    return stripes_copy(c, "ptr_image_pack_stripes", image, rank, 1u, world, stripe_rows, packed, false);
}

int ptr_image_unpack_stripes(ptr_ctx* c, ptr_handle image, uint32_t world, uint32_t stripe_rows, ptr_handle gathered) {
    return stripes_copy(c, "ptr_image_unpack_stripes", image, 0u, world, world, stripe_rows, gathered, true);
}

int ptr_set_sample_partition(ptr_ctx* c, uint32_t rank, uint32_t world) {
    if (!c) return PTR_E_INVALID;
    if (world == 0 || rank >= world) return fail(c, PTR_E_INVALID, "ptr_set_sample_partition: rank %u / world %u", rank, world);
    int rc = enter(c); if (rc) return rc;
    c->sample_rank = rank; c->sample_world = world;
    return PTR_OK;
}

int ptr_set_option(ptr_ctx* c, const char* key, int value) {
    if (!c || !key) return PTR_E_INVALID;
    // (a frame loop sets "dispatch.samples" before every dispatch: the same value again changes nothing)
    if (!(!strcmp(key, "dispatch.samples") && value == c->opt_dispatch_samples)) { int rc = enter(c); if (rc) return rc; }
    if (!strcmp(key, "extend.refill")) {
        if (value != 8 && value != 16 && value != 32 && value != 64) return fail(c, PTR_E_INVALID, "extend.refill: 8, 16, 32 or 64 idle lanes");
        c->opt_extend_refill = value;
        return PTR_OK;
    }
    if (!strcmp(key, "extend.leaf_weight")) {
        if (value != 1 && value != 2 && value != 4) return fail(c, PTR_E_INVALID, "extend.leaf_weight: 1, 2 or 4");
        c->opt_extend_leaf_weight = value;
        return PTR_OK;
    }
    if (!strcmp(key, "extend.alone_turns")) {
        if (value < 0 || value > 1000000) return fail(c, PTR_E_INVALID, "extend.alone_turns: 0..1000000");
        c->opt_extend_alone_turns = value;
        return PTR_OK;
    }
    if (!strcmp(key, "build.traversal_cost")) {
        if (value < 0 || value > 10000) return fail(c, PTR_E_INVALID, "build.traversal_cost: 0..10000 (percent of a triangle test)");
        c->opt_build_traversal_cost = value;
        return PTR_OK;
    }
    if (!strcmp(key, "build.cluster_radius")) {
        if (value < 0 || value > 64) return fail(c, PTR_E_INVALID, "build.cluster_radius: 0 (radix tree) or 1..64");
        c->opt_build_cluster_radius = value;
        return PTR_OK;
    }
    if (!strcmp(key, "extend.inner_reps")) {
        if (value < 1 || value > 3) return fail(c, PTR_E_INVALID, "extend.inner_reps: 1..3");
        c->opt_extend_inner_reps = value;
        return PTR_OK;
    }
    if (!strcmp(key, "extend.leaf_all")) {
        if (value != 0 && value != 1) return fail(c, PTR_E_INVALID, "extend.leaf_all: 0 or 1");
        c->opt_extend_leaf_all = value;
        return PTR_OK;
    }
    if (!strcmp(key, "profile.extend_only")) {
        if (value != 0 && value != 1) return fail(c, PTR_E_INVALID, "profile.extend_only: 0 or 1");
        c->opt_profile_extend_only = value;
        return PTR_OK;
    }
    if (!strcmp(key, "frames.in_flight")) {
        if (value < 1 || value > ptr_ctx::kMaxFrameSlots) return fail(c, PTR_E_INVALID, "frames.in_flight: 1..%d", ptr_ctx::kMaxFrameSlots);
        int rc = enter(c); if (rc) return rc;
        if ((rc = drain_frames(c))) return rc;
        c->opt_frames_in_flight = value;
        return PTR_OK;
    }
    if (!strcmp(key, "scene.index_words")) {
        if (value != 3 && value != 4) return fail(c, PTR_E_INVALID, "scene.index_words: 3 or 4");
        c->opt_index_words = value;
        return PTR_OK;
    }
    if (!strcmp(key, "scene.validate")) {
        if (value != 0 && value != 1) return fail(c, PTR_E_INVALID, "scene.validate: 0 or 1");
        c->opt_validate = value;
        return PTR_OK;
    }
    if (!strcmp(key, "bvh.auto_flatten")) {
        if (value != 0 && value != 1) return fail(c, PTR_E_INVALID, "bvh.auto_flatten: 0 or 1");
        c->opt_auto_flatten = value;
        return PTR_OK;
    }
    if (!strcmp(key, "dispatch.coalesce")) {
        if (value < 1 || value > 64) return fail(c, PTR_E_INVALID, "dispatch.coalesce: 1..64");
        c->opt_coalesce = value;
        return PTR_OK;
    }
    if (!strcmp(key, "dispatch.samples")) {
        if (value < 1 || value > 64) return fail(c, PTR_E_INVALID, "dispatch.samples: 1..64");
        c->opt_dispatch_samples = value;
        return PTR_OK;
    }
    if (!strcmp(key, "shade.pregather")) {
        if (value != 0 && value != 1) return fail(c, PTR_E_INVALID, "shade.pregather: 0 or 1");
        c->opt_shade_pregather = value;
        return PTR_OK;
    }
    if (!strcmp(key, "dispatch.coalesce_scratch_mb")) {
        if (value < 1) return fail(c, PTR_E_INVALID, "dispatch.coalesce_scratch_mb: at least 1");
        c->opt_coalesce_scratch_mb = value;
        return PTR_OK;
    }
    if (!strcmp(key, "debug.refuse_batches")) {
        c->opt_refuse_batches = value != 0;
        return PTR_OK;
    }
    if (!strcmp(key, "extend.blocks_per_cu")) {
        if (value < 1 || value > 64) return fail(c, PTR_E_INVALID, "extend.blocks_per_cu: 1..64");
        c->opt_extend_blocks_per_cu = value;
        return PTR_OK;
    }
    return fail(c, PTR_E_INVALID, "ptr_set_option: unknown key '%s'", key);
}

int ptr_set_capture(ptr_ctx* c, ptr_handle buffer) {
    if (!c) return PTR_E_INVALID;
    if (buffer != 0 && !find_buffer(c, buffer)) return fail(c, PTR_E_INVALID, "ptr_set_capture: unknown buffer %u", buffer);
    int rc = enter(c); if (rc) return rc;
    c->capture = buffer;
    return PTR_OK;
}

int ptr_stats_enable(ptr_ctx* c, int on) {
    if (!c) return PTR_E_INVALID;
    int rc = enter(c); if (rc) return rc;
    c->stats_on = on != 0;
    return PTR_OK;
}

int ptr_stats_read(ptr_ctx* c, uint64_t out[8], int reset) {
    if (!c || !out) return PTR_E_INVALID;
    int rc = enter(c); if (rc) return rc;
    unsigned long long host[8];
    PTR_HIP_CHECK(c, hipMemcpyAsync(host, c->d_stats, sizeof host, hipMemcpyDeviceToHost, c->stream));
    PTR_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < 8; i++) out[i] = host[i];
    out[4] = c->stats_frames;
    if (reset) {
        PTR_HIP_CHECK(c, hipMemsetAsync(c->d_stats, 0, sizeof host, c->stream));
        c->stats_frames = 0;
    }
    return PTR_OK;
}

int ptr_profile_enable(ptr_ctx* c, int on) {
    if (!c) return PTR_E_INVALID;
    int rc = enter(c); if (rc) return rc;
    c->profile_on = on != 0;
    return PTR_OK;
}

int ptr_profile_read(ptr_ctx* c, const char* name, uint64_t* launches, double* total_ms, int reset) {
    if (!c || !name) return PTR_E_INVALID;
    int rc = enter(c); if (rc) return rc;
    drain_profile(c);
    auto it = c->profile.find(name);
    if (launches) *launches = it == c->profile.end() ? 0 : it->second.launches;
    if (total_ms) *total_ms = it == c->profile.end() ? 0.0 : it->second.total_ms;
    if (reset && it != c->profile.end()) { it->second.launches = 0; it->second.total_ms = 0.0; }
    return PTR_OK;
}

int ptr_profile_names(ptr_ctx* c, char* dst, size_t size) {
    if (!c || !dst || size == 0) return PTR_E_INVALID;
    std::string s;
    for (auto& kv : c->profile) { if (!s.empty()) s += '\n'; s += kv.first; }
    if (s.size() + 1 > size) return fail(c, PTR_E_RANGE, "ptr_profile_names: need %zu bytes", s.size() + 1);
    memcpy(dst, s.c_str(), s.size() + 1);
    return PTR_OK;
}

}  // extern "C"
