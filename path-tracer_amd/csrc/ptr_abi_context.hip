// C ABI of the render hot path (include/ptr_hip.h): context, device resources, bindless tables, options, statistics, profile, partitions.
#include "ptr_abi_internal.h"

namespace ptr_abi {

namespace { thread_local std::string g_create_error; }
const char* create_error() { return g_create_error.c_str(); }

int fail(ptr_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->error = buf; else g_create_error = buf;
    return code;
}

int bind_device(ptr_ctx* c) {
    PTR_HIP_CHECK(c, hipSetDevice(c->device));
    return PTR_OK;
}
// Every entry point that looks at or changes what a dispatch sees starts here: held-back dispatches run first.
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
BufferRec* buffer_at(ptr_ctx* c, uint64_t addr, size_t need, ptr_handle* handle) {
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

}  // namespace ptr_abi

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
    if ((e = bind_static_scratch(c)) != hipSuccess) return bail("hipGetSymbolAddress(static scratch)", e);
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

const char* ptr_last_error(const ptr_ctx* c) { return c ? c->error.c_str() : create_error(); }

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
    for (auto& o : c->buffers) { if (o.flat == h) o.flat = 0; if (o.instances == h) o.instances = 0; if (o.spheres == h) { o.spheres = 0; o.sphere_count = 0; } }
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
