// C ABI of the render hot path (include/ptr_hip.h): the dispatch that turns the reference's 64-byte constant block into HIP launches, and the layouts it walks.
#include "ptr_abi_internal.h"

namespace ptr_abi {

// Host view of the 160-byte metadata the constant block points at.
// may_wait = false (ptr_dispatch_static): a metadata buffer without a host copy is refused instead of being copied back, which
// would make the caller wait for the stream
int read_meta(ptr_ctx* c, uint64_t addr, ptr::SceneMeta* out, bool may_wait) {
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
    uint64_t key[19] = {};
    for (int i = 1; i < 7; i++) { key[i - 1] = addr[i]; key[5 + i] = c->buffers[handles[i]].generation; }
    BufferRec* sphere_buf = wavefront && flat_buf && flat_buf->spheres ? find_buffer(c, flat_buf->spheres) : nullptr;
    key[16] = sphere_buf ? (uint64_t)flat_buf->spheres : 0;
    key[17] = sphere_buf ? sphere_buf->generation : 0;
    key[18] = sphere_buf ? flat_buf->sphere_count : 0;
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
                                                fv ? flat_buf->flat_triangles : 0u, top_buf ? top_buf->top_nodes : 0u, sphere_buf ? flat_buf->sphere_count : 0u, c->d_validate));
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
int dispatch_compute(ptr_ctx* c, const uint8_t constants[64], uint32_t gx, uint32_t gy, int force /*0 auto,1 mega,2 wavefront*/, bool no_alloc,
                     const ptr_ctx::Pending* batch) {
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
    if (flat_buf->spheres) {   // sphere leaves: the records they name
        const BufferRec* sb = find_buffer(c, flat_buf->spheres);
        if (!sb) return fail(c, PTR_E_STATE, "spheres: the buffer bound with ptr_bvh_bind_spheres is gone");
        if (top_buf) return fail(c, PTR_E_UNSUPPORTED, "spheres: a layout with sphere leaves cannot be instanced (ptr_bvh_bind_instances)");
        sv.spheres = (const ptr::Sphere*)sb->ptr;
        sv.sphere_count = flat_buf->sphere_count;
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
            PTR_HIP_CHECK(c, ptr::launch_wf_extend(ts, fv, w, b, sv.stats, c->opt_extend_refill, c->opt_extend_leaf_weight, c->opt_extend_leaf_all, c->opt_extend_inner_reps, b + 1 == meta.max_bounce && !sv.capture));
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
        { ScopedProfile prof(c, "wavefront.extend", ts); PTR_HIP_CHECK(c, ptr::launch_wf_extend(ts, fv, w, b, sv.stats, c->opt_extend_refill, c->opt_extend_leaf_weight, c->opt_extend_leaf_all, c->opt_extend_inner_reps, b + 1 == meta.max_bounce && !sv.capture)); }
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
        slot.reads[9] = flat_buf->spheres;
        c->next_slot = (c->next_slot + 1) % c->opt_frames_in_flight;
    }
    return PTR_OK;
}

}  // namespace ptr_abi

extern "C" {

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
        if (f->spheres) return fail(c, PTR_E_UNSUPPORTED, "ptr_bvh_bind_instances: this layout has sphere records bound (ptr_bvh_bind_spheres); a layout with sphere leaves cannot be instanced");
    }
    int rc = enter(c); if (rc) return rc;
    if ((rc = drain_frames(c))) return rc;
    f->instances = instances;
    return PTR_OK;
}

int ptr_bvh_bind_spheres(ptr_ctx* c, ptr_handle flat, ptr_handle spheres, uint32_t count) {
    if (!c) return PTR_E_INVALID;
    BufferRec* f = find_buffer(c, flat);
    if (!f) return fail(c, PTR_E_INVALID, "ptr_bvh_bind_spheres: unknown flat buffer %u", flat);
    if (spheres != 0) {
        BufferRec* s = find_buffer(c, spheres);
        if (!s) return fail(c, PTR_E_INVALID, "ptr_bvh_bind_spheres: unknown sphere buffer %u", spheres);
        if (count == 0 || count >= (1u << 29)) return fail(c, PTR_E_INVALID, "ptr_bvh_bind_spheres: 1 .. 2^29 - 1 spheres");
        if (s->size < (size_t)count * sizeof(ptr::Sphere)) return fail(c, PTR_E_RANGE, "ptr_bvh_bind_spheres: the buffer holds fewer than %u records of %zu bytes", count, sizeof(ptr::Sphere));
        if (spheres == flat) return fail(c, PTR_E_INVALID, "ptr_bvh_bind_spheres: the sphere buffer cannot be the flat buffer itself");
        if (f->instances) return fail(c, PTR_E_UNSUPPORTED, "ptr_bvh_bind_spheres: this layout has placed copies bound (ptr_bvh_bind_instances); a layout with sphere leaves cannot be instanced");
    }
    int rc = enter(c); if (rc) return rc;
    if ((rc = drain_frames(c))) return rc;
    f->spheres = spheres;
    f->sphere_count = spheres ? count : 0u;
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

}  // extern "C"
