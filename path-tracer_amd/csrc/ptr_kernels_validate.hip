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

// Threaded nodes (include/bvh.hpp:17-22): next_id in [-1, n), primitive_id in [-1, triangles), an inner node has a
// successor (its left child is the next record).  out[2] = violations.
__global__ __launch_bounds__(256) void validate_threaded(const ThreadedNode* nodes, uint32_t node_count, uint32_t triangle_count, uint32_t* out) {
    uint32_t bad = 0u;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < node_count; i += gridDim.x * 256u) {
        const int32_t next = nodes[i].next_id, prim = nodes[i].primitive_id;
        if (next < -1 || (next >= 0 && (uint32_t)next >= node_count)) bad++;
        if (prim < -1 || (prim >= 0 && (uint32_t)prim >= triangle_count)) bad++;
        if (prim == -1 && i + 1u >= node_count) bad++;
    }
    if (bad) atomicAdd(&out[2], bad);
}

// Flattened layout: child references inside the node / triangle arrays, primitive ids inside the index buffer.  out[3] = violations.
__global__ __launch_bounds__(256) void validate_flat(FlatView fv, uint32_t node_count, uint32_t flat_triangles, uint32_t triangle_count, uint32_t* out) {
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
        if (prim < 0 || (uint32_t)prim >= triangle_count) bad++;
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

}  // namespace

hipError_t launch_validate_scene(hipStream_t s, const SceneView& sv, uint32_t triangle_count, uint32_t threaded_nodes, const FlatView* fv,
                                 uint32_t flat_nodes, uint32_t flat_triangles, uint32_t top_nodes, uint32_t* out4) {
    hipError_t e = hipMemsetAsync(out4, 0, 16, s);
    if (e != hipSuccess) return e;
    auto grid = [](uint32_t n) { const uint32_t b = (n + 255u) / 256u; return dim3(b < 1u ? 1u : (b > 4096u ? 4096u : b)); };
    if (triangle_count) hipLaunchKernelGGL(validate_indices, grid(triangle_count), dim3(256), 0, s, sv, triangle_count, out4);
    if (threaded_nodes) hipLaunchKernelGGL(validate_threaded, grid(threaded_nodes), dim3(256), 0, s, sv.nodes, threaded_nodes, triangle_count, out4);
    if (fv) hipLaunchKernelGGL(validate_flat, grid(flat_nodes > flat_triangles ? flat_nodes : flat_triangles), dim3(256), 0, s, *fv, flat_nodes, flat_triangles, triangle_count, out4);
    if (fv && fv->top.nodes)
        hipLaunchKernelGGL(validate_top, grid(top_nodes > fv->top.count ? top_nodes : fv->top.count), dim3(256), 0, s, fv->top, top_nodes, flat_nodes, flat_triangles, out4);
    return hipGetLastError();
}

}  // namespace ptr
