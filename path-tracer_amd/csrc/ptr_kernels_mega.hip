// "compute" as a single kernel over the reference's threaded node array, and "tonemapping".
//
// This is the drop-in path: it consumes exactly the seven buffers + two images the reference's
// constant block names (compute.comp:109-120) and walks the 32-byte skip-link nodes the reference's
// builder emits, so every pixel follows the reference's traversal order, tie-breaking and RNG stream.
// One wave64 = one 8x8 workgroup of the reference's dispatch (compute.comp:14), one lane = one pixel.
#pragma clang fp contract(off)

#include "ptr_kernels.h"
#include "ptr_device_intersect.h"

namespace ptr {

// ray.h:165-193: stackless walk, left child first, leaves tested without a box test, a hit shrinks max_t,
// `t > max_t` rejects so an equal t replaces the earlier hit.
template <bool STATS>
PTR_DEV Hit hit_node(const SceneView& sv, const Ray& r, unsigned long long& n_nodes, unsigned long long& n_tris) {
    Hit best; best.prim = -1; best.t = 0.0f; best.u = 0.0f; best.v = 0.0f;
    float max_t = r.max_t;
    const f3 inv_dir = mk3(rcp(r.dir.x), rcp(r.dir.y), rcp(r.dir.z));
    const float4* nodes = (const float4*)sv.nodes;
    int32_t id = 0;
    while (id != -1) {
        const float4 lo = nodes[2 * id];      // min.xyz, next_id
        const float4 hi = nodes[2 * id + 1];  // max.xyz, primitive_id
        const int32_t next_id = __float_as_int(lo.w);
        const int32_t prim = __float_as_int(hi.w);
        if (STATS) n_nodes++;
        if (prim != -1) {
            float t, u, v;
            if (STATS) n_tris++;
            if (hit_triangle(sv, (uint32_t)prim, r, max_t, t, u, v)) {
                best.prim = prim; best.t = t; best.u = u; best.v = v;
                max_t = t;
            }
            id = next_id;
        } else if (hit_aabb(lo, hi, r.origin, inv_dir, r.min_t, max_t)) {
            id++;
        } else {
            id = next_id;
        }
    }
    return best;
}

// ray.h:80-103
PTR_DEV uint32_t hit_aabbs(const SceneView& sv, const Ray& r) {
    const f3 inv_dir = mk3(rcp(r.dir.x), rcp(r.dir.y), rcp(r.dir.z));
    const float4* nodes = (const float4*)sv.nodes;
    uint32_t count = 0;
    int32_t id = 0;
    while (id != -1) {
        const float4 lo = nodes[2 * id];
        const float4 hi = nodes[2 * id + 1];
        if (__float_as_int(hi.w) != -1) id = __float_as_int(lo.w);
        else if (hit_aabb(lo, hi, r.origin, inv_dir, r.min_t, r.max_t)) { count++; id++; }
        else id = __float_as_int(lo.w);
    }
    return count;
}

// compute.comp:264-289
template <bool STATS>
__global__ __launch_bounds__(64) void ptr_compute_megakernel(SceneView sv, uint32_t groups_x) {
    const SceneMeta& m = sv.meta;
    const uint32_t group = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const uint32_t gx = (group % groups_x) * 8u + (lane & 7u);
    const uint32_t gy = (group / groups_x) * 8u + (lane >> 3);
    const bool valid = gx < m.width && gy < m.height && owns_row(sv, gy);

    unsigned long long n_rays = 0, n_nodes = 0, n_tris = 0, n_shaded = 0;
    if (valid) {
        const int32_t cx = (int32_t)gx * m.downscale_factor, cy = (int32_t)gy * m.downscale_factor;
        uint32_t seed = pixel_seed(cx, cy, seed_sample_index(sv, m.sample_index));
        Ray r = generate_camera_ray(m, cx, cy, seed);

        Capture cap; cap.prim0 = -1; cap.t0 = 0.0f; cap.prim1 = -1; cap.t1 = 0.0f;
        f3 out_color = mk3(0.0f, 0.0f, 0.0f);
        if (m.debug_bvh == 1u) {
            const float n = (float)hit_aabbs(sv, r);
            out_color = mk3(n * 0.001f, n * 0.0f, n * 0.0f);
        } else {
            f3 throughput = mk3(1.0f, 1.0f, 1.0f);
            const uint32_t max_bounce = m.max_bounce, min_bounce = m.min_bounce;
            for (uint32_t bounce = 0; bounce < max_bounce; bounce++) {
                const Hit h = hit_node<STATS>(sv, r, n_nodes, n_tris);
                if (STATS) { n_rays++; if (h.prim >= 0) n_shaded++; }
                if (bounce == 0u) { cap.prim0 = h.prim; cap.t0 = h.prim >= 0 ? h.t : 0.0f; }
                if (bounce == 1u) { cap.prim1 = h.prim; cap.t1 = h.prim >= 0 ? h.t : 0.0f; }
                if (!shade_bounce(sv, bounce, min_bounce, h, r, throughput, seed, out_color)) break;
            }
        }
        if (sv.capture) sv.capture[(size_t)gy * m.width + gx] = cap;
        accumulate_pixel(sv.images[sv.accumulation_index], sv.images[sv.output_index], cx, cy, m.downscale_factor,
                         m.sample_index, out_color);
    }

    if (STATS && sv.stats) {  // every lane of the wave is here: reduce, then one atomic per counter
        for (int off = 32; off > 0; off >>= 1) {
            n_rays += __shfl_down(n_rays, off);
            n_nodes += __shfl_down(n_nodes, off);
            n_tris += __shfl_down(n_tris, off);
            n_shaded += __shfl_down(n_shaded, off);
        }
        if (lane == 0u) {
            atomicAdd(&sv.stats[0], n_rays); atomicAdd(&sv.stats[1], n_nodes);
            atomicAdd(&sv.stats[2], n_tris); atomicAdd(&sv.stats[3], n_shaded);
        }
    }
}

// tonemapping.comp:19-30.  The shader has no bounds guard; image_load/image_store apply the image rules
// (outside reads give 0, outside writes are dropped).
__global__ __launch_bounds__(64) void ptr_tonemapping(const SceneMeta* meta, const ImageDesc* images, uint32_t image_count,
                                                      uint32_t output_index, uint32_t accumulation_index, uint32_t groups_x) {
    const int32_t ds = meta->downscale_factor;
    const uint32_t group = blockIdx.x, lane = threadIdx.x;
    const int32_t cx = (int32_t)((group % groups_x) * 8u + (lane & 7u)) * ds;
    const int32_t cy = (int32_t)((group / groups_x) * 8u + (lane >> 3)) * ds;
    if (output_index >= image_count || accumulation_index >= image_count) return;
    const f4 a = image_load(images[accumulation_index], cx, cy);
    const f3 o = mk3(sqrtf(a.x), sqrtf(a.y), sqrtf(a.z));
    for (int32_t i = 0; i < ds; i++)
        for (int32_t j = 0; j < ds; j++) image_store(images[output_index], cx + i, cy + j, o);
}

// Sample-partition exchange helpers (no counterpart in the single-GPU reference).
//   linearize: out = (srgb_to_linear(acc.rgb), 1)   -- undo the accumulate kernel's sRGB encoding so that images add
//   resolve:   out = (linear_to_srgb(acc.rgb / acc.w), 1) -- after summing `acc.w` linearized images
__global__ __launch_bounds__(256) void ptr_linearize_resolve(const ImageDesc* images, uint32_t output_index, uint32_t accumulation_index,
                                                            uint32_t resolve) {
    const ImageDesc src = images[accumulation_index], dst = images[output_index];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= src.width * src.height || src.width != dst.width || src.height != dst.height) return;
    const float4 a = ((const float4*)src.data)[i];
    f3 o;
    if (resolve) o = linear_to_srgb(mk3(a.x / a.w, a.y / a.w, a.z / a.w));
    else o = srgb_to_linear(mk3(a.x, a.y, a.z));
    ((float4*)dst.data)[i] = make_float4(o.x, o.y, o.z, 1.0f);
}

hipError_t launch_linearize_resolve(hipStream_t s, const ImageDesc* images, uint32_t output_index, uint32_t accumulation_index,
                                    uint32_t pixels, bool resolve) {
    if (pixels == 0) return hipSuccess;
    hipLaunchKernelGGL(ptr_linearize_resolve, dim3((pixels + 255u) / 256u), dim3(256), 0, s, images, output_index, accumulation_index,
                       resolve ? 1u : 0u);
    return hipGetLastError();
}

// The lookup copy of an RGBA8 image (ImageDesc::footprints): entry (i, j) holds the 2 x 2 texels of the bilinear REPEAT lookup whose
// upper-left texel is (i, j).
__global__ __launch_bounds__(256) void ptr_footprints_rgba8(const uint32_t* linear, uint32_t* footprints, uint32_t width, uint32_t height) {
    const uint32_t i = blockIdx.x * 16u + (threadIdx.x & 15u), j = blockIdx.y * 16u + (threadIdx.x >> 4);
    if (i >= width || j >= height) return;
    const uint32_t i1 = i + 1u == width ? 0u : i + 1u, j1 = j + 1u == height ? 0u : j + 1u;
    uint4 e;
    e.x = linear[(size_t)j * width + i]; e.y = linear[(size_t)j * width + i1];
    e.z = linear[(size_t)j1 * width + i]; e.w = linear[(size_t)j1 * width + i1];
    ((uint4*)footprints)[footprint_entry((width + 1u) >> 1, i, j)] = e;
}
hipError_t launch_footprints_rgba8(hipStream_t s, const void* linear, void* footprints, uint32_t width, uint32_t height) {
    hipLaunchKernelGGL(ptr_footprints_rgba8, dim3((width + 15u) / 16u, (height + 15u) / 16u), dim3(256), 0, s, (const uint32_t*)linear, (uint32_t*)footprints, width, height);
    return hipGetLastError();
}

// What a present shows: the reference blits the RGBA32F output image to the swapchain's 8-bit UNORM image
// (vkapi::blit_full, src/vk-api.cpp:1194-1224; src/vk-renderer.cpp:199).  Float to UNORM as the API defines it: NaN gives 0, the
// value is clamped to [0, 1], scaled by 255 and rounded to the nearest integer.  One texel per lane, R in the low byte.
__global__ __launch_bounds__(256) void ptr_unorm8_pack(const float4* image, uint32_t texels, uint32_t* packed) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= texels) return;
    const float4 c = image[i];
    const float v[4] = { c.x, c.y, c.z, c.w };
    uint32_t word = 0u;
    for (int k = 0; k < 4; k++) {
        float f = v[k];
        f = f == f ? f : 0.0f;
        f = f < 0.0f ? 0.0f : (f > 1.0f ? 1.0f : f);
        word |= (uint32_t)rintf(f * 255.0f) << (8 * k);
    }
    packed[i] = word;
}
hipError_t launch_unorm8_pack(hipStream_t s, const void* image, uint32_t texels, void* packed) {
    if (texels == 0u) return hipSuccess;
    hipLaunchKernelGGL(ptr_unorm8_pack, dim3((texels + 255u) / 256u), dim3(256), 0, s, (const float4*)image, texels, (uint32_t*)packed);
    return hipGetLastError();
}

// Tile-partition exchange helpers (no counterpart in the single-GPU reference).  A frame of `height` rows is cut into
// stripes of `stripe_rows` rows; rank r of `world` owns the stripes s with s % world == r.  pack: the owned stripes of one
// rank, in order, as one contiguous chunk (`chunk_stripes` stripes, rows past the frame's end zero) -- what an all-gather
// sends.  unpack: `world` such chunks, [r][j] = stripe j * world + r, back into the rows of the frame.
__global__ __launch_bounds__(256) void ptr_stripes_copy(float4* image, uint32_t width, uint32_t height, uint32_t stripe_rows, uint32_t world,
                                                       uint32_t first_rank, uint32_t ranks, uint32_t chunk_stripes, float4* packed, uint32_t unpack) {
    const uint64_t per_stripe = (uint64_t)stripe_rows * width;
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;          // texel of the packed side
    if (i >= (uint64_t)ranks * chunk_stripes * per_stripe) return;
    const uint32_t r = first_rank + (uint32_t)(i / (chunk_stripes * per_stripe));
    const uint64_t in_chunk = i % (chunk_stripes * per_stripe);
    const uint32_t j = (uint32_t)(in_chunk / per_stripe);
    const uint64_t in_stripe = in_chunk % per_stripe;
    const uint64_t row = (uint64_t)(j * world + r) * stripe_rows + in_stripe / width;
    const bool inside = row < height;
    if (unpack) { if (inside) image[row * width + in_stripe % width] = packed[i]; }
    else packed[i] = inside ? image[row * width + in_stripe % width] : make_float4(0.f, 0.f, 0.f, 0.f);
}

hipError_t launch_stripes_copy(hipStream_t s, float4* image, uint32_t width, uint32_t height, uint32_t stripe_rows, uint32_t world,
                               uint32_t first_rank, uint32_t ranks, float4* packed, bool unpack) {
    const uint32_t stripes = (height + stripe_rows - 1u) / stripe_rows;
    const uint32_t chunk_stripes = (stripes + world - 1u) / world;
    const uint64_t texels = (uint64_t)ranks * chunk_stripes * stripe_rows * width;
    if (texels == 0) return hipSuccess;
    hipLaunchKernelGGL(ptr_stripes_copy, dim3((uint32_t)((texels + 255u) / 256u)), dim3(256), 0, s, image, width, height, stripe_rows, world,
                       first_rank, ranks, chunk_stripes, packed, unpack ? 1u : 0u);
    return hipGetLastError();
}

hipError_t launch_megakernel(hipStream_t s, const SceneView& sv, uint32_t groups_x, uint32_t groups_y, bool stats) {
    const uint32_t blocks = groups_x * groups_y;
    if (blocks == 0) return hipSuccess;
    if (stats) hipLaunchKernelGGL(ptr_compute_megakernel<true>, dim3(blocks), dim3(64), 0, s, sv, groups_x);
    else hipLaunchKernelGGL(ptr_compute_megakernel<false>, dim3(blocks), dim3(64), 0, s, sv, groups_x);
    return hipGetLastError();
}

hipError_t launch_tonemap(hipStream_t s, const SceneMeta* meta, const ImageDesc* images, uint32_t image_count,
                          uint32_t output_index, uint32_t accumulation_index, uint32_t groups_x, uint32_t groups_y) {
    const uint32_t blocks = groups_x * groups_y;
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(ptr_tonemapping, dim3(blocks), dim3(64), 0, s, meta, images, image_count, output_index, accumulation_index, groups_x);
    return hipGetLastError();
}

}  // namespace ptr
