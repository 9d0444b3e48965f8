// C ABI of the render hot path (include/ptr_hip.h): "dispatch.coalesce" -- single-sample dispatches held back and rendered as one batched dispatch.
#include "ptr_abi_internal.h"

namespace ptr_abi {

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
        p.reads[9] = flat ? flat->spheres : 0;
    }
    if (p.count >= (uint32_t)c->opt_coalesce) return flush_pending(c);
    return PTR_OK;
}

}  // namespace ptr_abi

extern "C" {

int ptr_flush(ptr_ctx* c) {
    if (!c) return PTR_E_INVALID;
    return enter(c);
}

}  // extern "C"
