// "compute" as a wavefront pipeline over the flattened BVH layout:
//
//   generate   one lane per pixel: RNG seed + thin-lens camera ray -> ray/state queues (SoA)
//   extend     persistent wave64 traversal: waves pull 64 rays at a time from the queue with one
//              atomic, walk the 64-byte two-child nodes near-child-first with a per-lane stack in LDS
//              (overflow in global memory), test the 64-byte pre-gathered triangle records
//   shade      one lane per live path: material fetch, BRDF sampling, Russian roulette; survivors are
//              compacted into the next generation's queues (ballot + one atomic per wave)
//   accumulate one lane per pixel: temporal mean with the accumulation image -> output image
//
// Queues are structure-of-arrays so that every load/store of a wave is one contiguous 256-byte run.
// Results do not depend on queue order: a path carries its pixel index, RNG state and throughput.
#pragma clang fp contract(off)

#include "ptr_kernels.h"

namespace ptr {

constexpr float kBoxPad = 1.0000002384185791015625f;  // 1 + 2^-22: boxes cull conservatively

// Appends the lanes with `keep` set to a queue: returns the slot for this lane (undefined when !keep).
PTR_DEV uint32_t wave_append(uint32_t* counter, bool keep) {
    const unsigned long long mask = __ballot(keep);
    if (mask == 0ull) return 0u;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t leader = (uint32_t)__builtin_ctzll(mask);
    uint32_t base = 0u;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, (int)leader);
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// ---- generate ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ptr_wf_generate(SceneView sv, WavefrontScratch w, uint32_t domain_w, uint32_t domain_h,
                                                      uint32_t tiles_x) {
    const SceneMeta& m = *sv.meta;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);  // one 8x8 tile per wave
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gx = (wave % tiles_x) * 8u + (lane & 7u);
    const uint32_t gy = (wave / tiles_x) * 8u + (lane >> 3);
    const bool valid = gx < domain_w && gy < domain_h && owns_row(sv, gy);

    Ray r; r.origin = mk3(0, 0, 0); r.dir = mk3(0, 0, 0);
    uint32_t seed = 0u;
    if (valid) {
        const int32_t cx = (int32_t)gx * m.downscale_factor, cy = (int32_t)gy * m.downscale_factor;
        seed = pixel_seed(cx, cy, seed_sample_index(sv, m.sample_index));
        r = generate_camera_ray(m, cx, cy, seed);  // seed by value: the bounce loop restarts from it
        if (sv.capture) { Capture c; c.prim0 = -1; c.t0 = 0.0f; c.prim1 = -1; c.t1 = 0.0f; sv.capture[(size_t)gy * m.width + gx] = c; }
    }
    const uint32_t slot = wave_append(&w.counters[0], valid);
    if (!valid) return;
    w.ray[0][0][slot] = r.origin.x; w.ray[0][1][slot] = r.origin.y; w.ray[0][2][slot] = r.origin.z;
    w.ray[0][3][slot] = r.dir.x; w.ray[0][4][slot] = r.dir.y; w.ray[0][5][slot] = r.dir.z;
    w.throughput[0][0][slot] = 1.0f; w.throughput[0][1][slot] = 1.0f; w.throughput[0][2][slot] = 1.0f;
    w.seed[0][slot] = seed;
    w.pixel[0][slot] = gy * m.width + gx;
}

// ---- extend --------------------------------------------------------------------------------------------
struct SlabResult { bool hit; float entry; };

PTR_DEV SlabResult slab(const float4 lo, const float4 hi, const f3 o, const f3 inv, float min_t, float best_t) {
    const f3 fv = (mk3(hi.x, hi.y, hi.z) - o) * inv;
    const f3 nv = (mk3(lo.x, lo.y, lo.z) - o) * inv;
    const float t1 = min_num(min_num(max_num(fv.x, nv.x), min_num(max_num(fv.y, nv.y), max_num(fv.z, nv.z))), best_t);
    const float t0 = max_num(max_num(max_num(min_num(fv.x, nv.x), max_num(min_num(fv.y, nv.y), min_num(fv.z, nv.z))), min_t), 0.0f);
    SlabResult s;
    s.entry = t0;
    s.hit = t1 * kBoxPad >= t0;
    return s;
}

template <bool STATS>
__global__ __launch_bounds__(kExtendThreads, 4) void ptr_wf_extend(FlatView fv, WavefrontScratch w, uint32_t gen, uint32_t slot_b,
                                                                unsigned long long* stats) {
    __shared__ int32_t lds_stack[(kExtendThreads / 64) * kStackLds * 64];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_in_block = threadIdx.x >> 6;
    int32_t* my_lds = lds_stack + wave_in_block * kStackLds * 64 + lane;
    int32_t* my_spill = w.stack_spill + (size_t)(blockIdx.x * (kExtendThreads / 64) + wave_in_block) * kStackSpill * 64 + lane;

    const uint32_t count = w.counters[slot_b];
    uint32_t* cursor = &w.counters[kMaxBounceSlots + slot_b];
    unsigned long long n_rays = 0, n_nodes = 0, n_tris = 0;

    for (;;) {
        uint32_t base = 0u;
        if (lane == 0u) base = atomicAdd(cursor, 64u);
        base = __shfl(base, 0);
        if (base >= count) break;
        const uint32_t slot = base + lane;
        if (slot < count) {
            const f3 o = mk3(w.ray[gen][0][slot], w.ray[gen][1][slot], w.ray[gen][2][slot]);
            const f3 d = mk3(w.ray[gen][3][slot], w.ray[gen][4][slot], w.ray[gen][5][slot]);
            const f3 inv = mk3(rcp(d.x), rcp(d.y), rcp(d.z));
            const float min_t = 0.001f;
            float best_t = 1e15f, best_u = 0.0f, best_v = 0.0f;
            int32_t best_rank = -1, best_prim = -1;
            int32_t cur = fv.root;
            uint32_t sp = 0u;
            if (STATS) n_rays++;
            for (;;) {
                if (cur >= 0) {
                    const float4 q0 = fv.nodes[4 * cur], q1 = fv.nodes[4 * cur + 1];
                    const float4 q2 = fv.nodes[4 * cur + 2], q3 = fv.nodes[4 * cur + 3];
                    if (STATS) n_nodes++;
                    const SlabResult l = slab(q0, q1, o, inv, min_t, best_t);
                    const SlabResult r = slab(q2, q3, o, inv, min_t, best_t);
                    const int32_t left = __float_as_int(q0.w), right = __float_as_int(q1.w);
                    if (l.hit && r.hit) {
                        const bool right_first = r.entry < l.entry;
                        const int32_t far_ref = right_first ? left : right;
                        if (sp < kStackLds) my_lds[sp * 64u] = far_ref;
                        else my_spill[(sp - kStackLds) * 64u] = far_ref;
                        sp++;
                        cur = right_first ? right : left;
                        continue;
                    }
                    if (l.hit) { cur = left; continue; }
                    if (r.hit) { cur = right; continue; }
                } else {
                    const uint32_t ref = (uint32_t)~cur;
                    const uint32_t first = ref >> 2, tri_count = (ref & 3u) + 1u;
                    for (uint32_t k = 0; k < tri_count; k++) {
                        const float4* tr = fv.triangles + 4 * (size_t)(first + k);
                        const float4 a = tr[0], b = tr[1], c = tr[2], n4 = tr[3];
                        if (STATS) n_tris++;
                        const f3 e1 = mk3(b.x, b.y, b.z), e2 = mk3(c.x, c.y, c.z), n = mk3(n4.x, n4.y, n4.z);
                        const f3 ov = o - mk3(a.x, a.y, a.z);
                        const f3 q = cross(ov, d);
                        const float dd = rcp(dot(d, n));
                        const float u = dd * dot(-q, e2);
                        const float v = dd * dot(q, e1);
                        const float t = dd * dot(-n, ov);
                        if (u < 0.0f || u > 1.0f || v < 0.0f || u + v > 1.0f || t < min_t || t > best_t) continue;
                        if (t != t || u != u || v != v) continue;
                        const int32_t rank = __float_as_int(a.w);
                        if (best_prim >= 0 && t == best_t && rank < best_rank) continue;
                        best_t = t; best_u = u; best_v = v; best_rank = rank; best_prim = __float_as_int(b.w);
                    }
                }
                if (sp == 0u) break;
                sp--;
                cur = sp < kStackLds ? my_lds[sp * 64u] : my_spill[(sp - kStackLds) * 64u];
            }
            w.hits[slot] = make_float4(best_t, best_u, best_v, __int_as_float(best_prim));
        }
    }
    if (STATS && stats) {
        for (int off = 32; off > 0; off >>= 1) {
            n_rays += __shfl_down(n_rays, off);
            n_nodes += __shfl_down(n_nodes, off);
            n_tris += __shfl_down(n_tris, off);
        }
        if (lane == 0u) { atomicAdd(&stats[0], n_rays); atomicAdd(&stats[1], n_nodes); atomicAdd(&stats[2], n_tris); }
    }
}

// ---- shade ---------------------------------------------------------------------------------------------
template <bool STATS>
__global__ __launch_bounds__(256) void ptr_wf_shade(SceneView sv, WavefrontScratch w, uint32_t gen, uint32_t slot_b, uint32_t slot_next,
                                                   uint32_t bounce, uint32_t max_bounce, uint32_t min_bounce) {
    const uint32_t count = w.counters[slot_b];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t wave_base = i & ~63u;
    if (wave_base >= count) return;  // whole wave idle
    const bool live = i < count;

    bool go_on = false;
    Ray r; r.origin = mk3(0, 0, 0); r.dir = mk3(0, 0, 0); r.min_t = 0.001f; r.max_t = 1e15f;
    f3 throughput = mk3(0, 0, 0), result = mk3(0, 0, 0);
    uint32_t seed = 0u, pixel = 0u;
    unsigned long long shaded = 0;
    if (live) {
        r.origin = mk3(w.ray[gen][0][i], w.ray[gen][1][i], w.ray[gen][2][i]);
        r.dir = mk3(w.ray[gen][3][i], w.ray[gen][4][i], w.ray[gen][5][i]);
        throughput = mk3(w.throughput[gen][0][i], w.throughput[gen][1][i], w.throughput[gen][2][i]);
        seed = w.seed[gen][i];
        pixel = w.pixel[gen][i];
        const float4 h4 = w.hits[i];
        Hit h; h.t = h4.x; h.u = h4.y; h.v = h4.z; h.prim = __float_as_int(h4.w);
        if (STATS && h.prim >= 0) shaded = 1;
        if (sv.capture && bounce < 2u) {
            Capture* c = &sv.capture[pixel];
            if (bounce == 0u) { c->prim0 = h.prim; c->t0 = h.prim >= 0 ? h.t : 0.0f; }
            else { c->prim1 = h.prim; c->t1 = h.prim >= 0 ? h.t : 0.0f; }
        }
        go_on = shade_bounce(sv, bounce, min_bounce, h, r, throughput, seed, result);
        if (go_on && bounce + 1u >= max_bounce) { go_on = false; result = mk3(0.0f, 0.0f, 0.0f); }  // loop exhausted: radiance
        if (!go_on) w.samples[pixel] = make_float4(result.x, result.y, result.z, 1.0f);
    }
    const uint32_t next = gen ^ 1u;
    const uint32_t slot = wave_append(&w.counters[slot_next], go_on);
    if (go_on) {
        w.ray[next][0][slot] = r.origin.x; w.ray[next][1][slot] = r.origin.y; w.ray[next][2][slot] = r.origin.z;
        w.ray[next][3][slot] = r.dir.x; w.ray[next][4][slot] = r.dir.y; w.ray[next][5][slot] = r.dir.z;
        w.throughput[next][0][slot] = throughput.x; w.throughput[next][1][slot] = throughput.y; w.throughput[next][2][slot] = throughput.z;
        w.seed[next][slot] = seed;
        w.pixel[next][slot] = pixel;
    }
    if (STATS && sv.stats) {
        for (int off = 32; off > 0; off >>= 1) shaded += __shfl_down(shaded, off);
        if ((threadIdx.x & 63u) == 0u && shaded) atomicAdd(&sv.stats[3], shaded);
    }
}

// ---- accumulate ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ptr_wf_accumulate(SceneView sv, WavefrontScratch w, uint32_t domain_w, uint32_t domain_h,
                                                        uint32_t tiles_x) {
    const SceneMeta& m = *sv.meta;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gx = (wave % tiles_x) * 8u + (lane & 7u);
    const uint32_t gy = (wave / tiles_x) * 8u + (lane >> 3);
    if (gx >= domain_w || gy >= domain_h || !owns_row(sv, gy)) return;
    const float4 s = w.samples[gy * m.width + gx];
    accumulate_pixel(sv.images[sv.accumulation_index], sv.images[sv.output_index], (int32_t)gx * m.downscale_factor,
                     (int32_t)gy * m.downscale_factor, m.downscale_factor, m.sample_index, mk3(s.x, s.y, s.z));
}

// ---- launch wrappers -----------------------------------------------------------------------------------
static inline uint32_t tile_blocks(uint32_t domain_w, uint32_t domain_h, uint32_t& tiles_x) {
    tiles_x = (domain_w + 7u) / 8u;
    const uint32_t tiles = tiles_x * ((domain_h + 7u) / 8u);
    return (tiles + 3u) / 4u;
}

hipError_t launch_wf_generate(hipStream_t s, const SceneView& sv, const WavefrontScratch& w, uint32_t domain_w, uint32_t domain_h) {
    uint32_t tiles_x;
    const uint32_t blocks = tile_blocks(domain_w, domain_h, tiles_x);
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(ptr_wf_generate, dim3(blocks), dim3(256), 0, s, sv, w, domain_w, domain_h, tiles_x);
    return hipGetLastError();
}

hipError_t launch_wf_extend(hipStream_t s, const FlatView& fv, const WavefrontScratch& w, uint32_t bounce, unsigned long long* stats) {
    const uint32_t slot_b = bounce % kMaxBounceSlots;
    if (stats) hipLaunchKernelGGL(ptr_wf_extend<true>, dim3(w.extend_blocks), dim3(kExtendThreads), 0, s, fv, w, bounce & 1u, slot_b, stats);
    else hipLaunchKernelGGL(ptr_wf_extend<false>, dim3(w.extend_blocks), dim3(kExtendThreads), 0, s, fv, w, bounce & 1u, slot_b, stats);
    return hipGetLastError();
}

hipError_t launch_wf_shade(hipStream_t s, const SceneView& sv, const WavefrontScratch& w, uint32_t bounce, uint32_t max_bounce,
                           uint32_t min_bounce, uint32_t domain_w, uint32_t domain_h) {
    const uint32_t upper = domain_w * domain_h;  // live paths never exceed the pixel count
    const uint32_t blocks = (upper + 255u) / 256u;
    if (blocks == 0) return hipSuccess;
    const uint32_t slot_b = bounce % kMaxBounceSlots, slot_next = (bounce + 1u) % kMaxBounceSlots;
    if (sv.stats) hipLaunchKernelGGL(ptr_wf_shade<true>, dim3(blocks), dim3(256), 0, s, sv, w, bounce & 1u, slot_b, slot_next, bounce, max_bounce, min_bounce);
    else hipLaunchKernelGGL(ptr_wf_shade<false>, dim3(blocks), dim3(256), 0, s, sv, w, bounce & 1u, slot_b, slot_next, bounce, max_bounce, min_bounce);
    return hipGetLastError();
}

hipError_t launch_wf_accumulate(hipStream_t s, const SceneView& sv, const WavefrontScratch& w, uint32_t domain_w, uint32_t domain_h) {
    uint32_t tiles_x;
    const uint32_t blocks = tile_blocks(domain_w, domain_h, tiles_x);
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(ptr_wf_accumulate, dim3(blocks), dim3(256), 0, s, sv, w, domain_w, domain_h, tiles_x);
    return hipGetLastError();
}

}  // namespace ptr
