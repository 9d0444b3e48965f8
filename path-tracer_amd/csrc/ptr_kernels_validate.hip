// Validation of what a "compute" dispatch is about to dereference.  The constant block only names addresses; the
// CONTENTS of the index buffer and of the two tree layouts are indices into other buffers, and an out-of-range one
// would be a wild GPU access.  These kernels reduce each buffer to a few maxima / violation counts that the host
// compares with the sizes of the buffers they point into, once per upload (the ABI caches the verdict by the
// buffers' write generations).
#include "ptr_kernels.h"

namespace ptr {

namespace {

// out[0] = max vertex id, out[1] = max material id
__global__ __launch_bounds__(256) void validate_indices(SceneView sv, uint32_t triangle_count, uint32_t* out) {
    __shared__ uint32_t red[2];
    if (threadIdx.x < 2) red[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t max_vertex = 0u, max_material = 0u;
    for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < triangle_count; t += gridDim.x * 256u) {
        uint32_t i0, i1, i2, mat;
        decode_triangle(sv, t, i0, i1, i2, mat);
        const uint32_t m = i0 > i1 ? (i0 > i2 ? i0 : i2) : (i1 > i2 ? i1 : i2);
        max_vertex = m > max_vertex ? m : max_vertex;
        max_material = mat > max_material ? mat : max_material;
    }
    atomicMax(&red[0], max_vertex); atomicMax(&red[1], max_material);
    __syncthreads();
    if (threadIdx.x < 2) atomicMax(&out[threadIdx.x], red[threadIdx.x]);
}

// Threaded nodes (include/bvh.hpp:17-22): next_id is -1 or a LATER record (the builder numbers nodes depth first,
// src/bvh.cpp:264-279, so a skip link always points forward; with that, every step of the stackless walk moves to a higher
// index and the walk ends after at most n steps -- a backward or self link would loop forever), primitive_id in
// [-1, triangles), an inner node has a successor (its left child is the next record).  out[2] = violations.
__global__ __launch_bounds__(256) void validate_threaded(const ThreadedNode* nodes, uint32_t node_count, uint32_t triangle_count, uint32_t* out) {
    uint32_t bad = 0u;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < node_count; i += gridDim.x * 256u) {
        const int32_t next = nodes[i].next_id, prim = nodes[i].primitive_id;
        if (next < -1 || (next >= 0 && ((uint32_t)next >= node_count || (uint32_t)next <= i))) bad++;
        if (prim < -1 || (prim >= 0 && (uint32_t)prim >= triangle_count)) bad++;
        if (prim == -1 && i + 1u >= node_count) bad++;
    }
    if (bad) atomicAdd(&out[2], bad);
}

// Flattened layout: child references inside the node / triangle arrays, primitive ids inside the index buffer.  out[3] = violations.
__global__ __launch_bounds__(256) void validate_flat(FlatView fv, uint32_t node_count, uint32_t flat_triangles, uint32_t triangle_count, uint32_t sphere_count,
                                                    uint32_t* out) {
    uint32_t bad = 0u;
    const uint32_t stride = gridDim.x * 256u;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < node_count; i += stride) {
        const int4 refs = ((const int4*)fv.nodes)[4 * (size_t)i + 1];
        const int32_t r[4] = { refs.x, refs.y, refs.z, refs.w };
        for (int k = 0; k < 4; k++) {
            if (r[k] == INT32_MIN) { if (k < 2) bad++; continue; }          // the first two children always exist
            if (r[k] >= 0) { if ((uint32_t)r[k] >= node_count) bad++; }
            else {
                const uint32_t ref = (uint32_t)~r[k];
                if ((uint64_t)(ref >> 2) + (ref & 3u) + 1u > flat_triangles) bad++;
            }
        }
    }
    for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < flat_triangles; t += stride) {
        const int32_t prim = __float_as_int(fv.triangles[4 * (size_t)t + 1].w);
        const int32_t kind = __float_as_int(fv.triangles[4 * (size_t)t + 2].w);
        if (kind == 1) {   // a sphere record names one of the bound spheres
            if (prim < 0 || (prim & kSpherePrimTag) == 0 || (uint32_t)(prim & ~kSpherePrimTag) >= sphere_count) bad++;
        } else if (kind != 0 || prim < 0 || (uint32_t)prim >= triangle_count) {
            bad++;
        }
    }
    if (bad) atomicAdd(&out[3], bad);
}

// Top level over placed copies: inner references inside its node array, every leaf an instance that exists, every
// instance's object root inside the object's arrays.  out[3] += violations.
__global__ __launch_bounds__(256) void validate_top(InstanceView top, uint32_t top_nodes, uint32_t object_nodes, uint32_t object_triangles, uint32_t* out) {
    uint32_t bad = 0u;
    const uint32_t stride = gridDim.x * 256u;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < top_nodes; i += stride) {
        const int4 refs = ((const int4*)top.nodes)[4 * (size_t)i + 1];
        const int32_t r[4] = { refs.x, refs.y, refs.z, refs.w };
        for (int k = 0; k < 4; k++) {
            if (r[k] == INT32_MIN) { if (k < 2) bad++; continue; }
            if (r[k] >= 0) { if ((uint32_t)r[k] >= top_nodes) bad++; }
            else {
                const uint32_t ref = (uint32_t)~r[k];
                if ((ref & kInstanceTag) == 0u || (ref & ~kInstanceTag) >= top.count) bad++;
            }
        }
    }
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < top.count; i += stride) {
        const int32_t root = __float_as_int(top.records[4 * (size_t)i + 3].x);
        if (root >= 0) { if ((uint32_t)root >= object_nodes) bad++; }
        else {
            const uint32_t ref = (uint32_t)~root;
            if ((uint64_t)(ref >> 2) + (ref & 3u) + 1u > object_triangles) bad++;
        }
    }
    if (bad) atomicAdd(&out[3], bad);
}

// One level of the tree, breadth first: every frontier node hands its inner children on, each with the number of
// references that can be pending on a depth-first walk's stack when it is entered (the parent's figure plus the parent's
// other children).  counters: [0] next frontier length, [1] largest figure seen, [2] frontier overflow (a node reached
// along more than one path: not a tree).  One atomic per wave.
struct DepthItem { uint32_t node, pending; };
__global__ __launch_bounds__(256) void depth_level(const float4* nodes, uint32_t node_count, const DepthItem* in, uint32_t count, DepthItem* out,
                                                   uint32_t capacity, uint32_t* counters) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    int32_t r[4] = { INT32_MIN, INT32_MIN, INT32_MIN, INT32_MIN };
    uint32_t pending = 0u, inner = 0u;
    if (i < count) {
        const DepthItem item = in[i];
        const int4 refs = ((const int4*)nodes)[4 * (size_t)item.node + 1];
        r[0] = refs.x; r[1] = refs.y; r[2] = refs.z; r[3] = refs.w;
        uint32_t children = 0u;
        for (int k = 0; k < 4; k++) {
            if (r[k] == INT32_MIN) continue;
            children++;
            if (r[k] >= 0 && (uint32_t)r[k] < node_count) inner++; else if (r[k] >= 0) r[k] = INT32_MIN;
        }
        pending = item.pending + (children ? children - 1u : 0u);
        atomicMax(&counters[1], pending);
    }
    // wave-wide exclusive prefix of `inner` (0..4): three ballots, one per bit
    const unsigned long long b0 = __ballot((inner & 1u) != 0u), b1 = __ballot((inner & 2u) != 0u), b2 = __ballot((inner & 4u) != 0u);
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t offset = (uint32_t)__popcll(b0 & below) + 2u * (uint32_t)__popcll(b1 & below) + 4u * (uint32_t)__popcll(b2 & below);
    const uint32_t total = (uint32_t)__popcll(b0) + 2u * (uint32_t)__popcll(b1) + 4u * (uint32_t)__popcll(b2);
    uint32_t base = 0u;
    if (lane == 0u && total) base = atomicAdd(&counters[0], total);
    base = __shfl(base, 0);
    uint32_t at = base + offset;
    for (int k = 0; k < 4; k++) {
        if (r[k] < 0) continue;
        if (at < capacity) { DepthItem next; next.node = (uint32_t)r[k]; next.pending = pending; out[at] = next; }
        else counters[2] = 1u;
        at++;
    }
}

}  // namespace

hipError_t measure_stack_need(hipStream_t s, const float4* nodes, uint32_t node_count, int32_t root, uint32_t limit, uint32_t* need, int* status) {
    *need = 0u; *status = 0;
    if (root < 0 || node_count == 0u) return hipSuccess;     // the root is a leaf: nothing is ever pending
    if ((uint32_t)root >= node_count) { *status = -1; return hipSuccess; }
    DepthItem* frontier[2] = { nullptr, nullptr };
    uint32_t* counters = nullptr;
    auto cleanup = [&]() { (void)hipFree(frontier[0]); (void)hipFree(frontier[1]); (void)hipFree(counters); };
    hipError_t e;
    if ((e = hipMalloc((void**)&frontier[0], (size_t)node_count * sizeof(DepthItem))) != hipSuccess) { cleanup(); return e; }
    if ((e = hipMalloc((void**)&frontier[1], (size_t)node_count * sizeof(DepthItem))) != hipSuccess) { cleanup(); return e; }
    if ((e = hipMalloc((void**)&counters, 16)) != hipSuccess) { cleanup(); return e; }
    const DepthItem first{ (uint32_t)root, 0u };
    uint32_t host[4] = { 0u, 0u, 0u, 0u };
    if ((e = hipMemcpyAsync(frontier[0], &first, sizeof first, hipMemcpyHostToDevice, s)) != hipSuccess) { cleanup(); return e; }
    if ((e = hipMemcpyAsync(counters, host, sizeof host, hipMemcpyHostToDevice, s)) != hipSuccess) { cleanup(); return e; }
    uint32_t count = 1u;
    int cur = 0;
    // every level adds at least one pending reference, so a tree within the limit has at most limit + 1 levels
    for (uint32_t level = 0; count > 0u; level++) {
        if (level > limit + 1u) { *status = -2; break; }
        if ((e = hipMemsetAsync(counters, 0, 4, s)) != hipSuccess) { cleanup(); return e; }
        hipLaunchKernelGGL(depth_level, dim3((count + 255u) / 256u), dim3(256), 0, s, nodes, node_count, frontier[cur], count, frontier[cur ^ 1], node_count, counters);
        if ((e = hipGetLastError()) != hipSuccess) { cleanup(); return e; }
        if ((e = hipMemcpyAsync(host, counters, sizeof host, hipMemcpyDeviceToHost, s)) != hipSuccess) { cleanup(); return e; }
        if ((e = hipStreamSynchronize(s)) != hipSuccess) { cleanup(); return e; }
        if (host[2] != 0u || host[0] > node_count) { *status = -3; break; }
        if (host[1] > limit) { *status = -2; break; }
        count = host[0];
        cur ^= 1;
    }
    *need = host[1];
    cleanup();
    return hipSuccess;
}

hipError_t launch_validate_scene(hipStream_t s, const SceneView& sv, uint32_t triangle_count, uint32_t threaded_nodes, const FlatView* fv,
                                 uint32_t flat_nodes, uint32_t flat_triangles, uint32_t top_nodes, uint32_t sphere_count, uint32_t* out4) {
    hipError_t e = hipMemsetAsync(out4, 0, 16, s);
    if (e != hipSuccess) return e;
    auto grid = [](uint32_t n) { const uint32_t b = (n + 255u) / 256u; return dim3(b < 1u ? 1u : (b > 4096u ? 4096u : b)); };
    if (triangle_count) hipLaunchKernelGGL(validate_indices, grid(triangle_count), dim3(256), 0, s, sv, triangle_count, out4);
    if (threaded_nodes) hipLaunchKernelGGL(validate_threaded, grid(threaded_nodes), dim3(256), 0, s, sv.nodes, threaded_nodes, triangle_count, out4);
    if (fv) hipLaunchKernelGGL(validate_flat, grid(flat_nodes > flat_triangles ? flat_nodes : flat_triangles), dim3(256), 0, s, *fv, flat_nodes, flat_triangles, triangle_count, sphere_count, out4);
    if (fv && fv->top.nodes)
        hipLaunchKernelGGL(validate_top, grid(top_nodes > fv->top.count ? top_nodes : fv->top.count), dim3(256), 0, s, fv->top, top_nodes, flat_nodes, flat_triangles, out4);
    return hipGetLastError();
}

}  // namespace ptr
