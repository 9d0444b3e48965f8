// "compute" as a wavefront pipeline over the flattened BVH layout:
//
//   generate   one lane per pixel: RNG seed + thin-lens camera ray -> ray/state queues (SoA)
//   extend     persistent wave64 traversal of the 4-wide, 8-bit-quantised 64-byte nodes, nearest child
//              first, per-lane stack in LDS (overflow in global memory), 64-byte pre-gathered triangles
//   shade      one lane per live path: material fetch, BRDF sampling, Russian roulette; survivors are
//              compacted into the next generation's queues (ballot + one atomic per wave)
//   accumulate one lane per pixel: temporal mean with the accumulation image -> output image
//
// Queues are arrays of 16-byte records so that every load/store of a wave is one contiguous 1 KiB run.
// Results do not depend on queue order: a path carries its pixel index, RNG state and throughput.
#pragma clang fp contract(off)

#include "ptr_kernels.h"
#include "ptr_device_intersect.h"

namespace ptr {

constexpr float kBoxPad = 1.0000002384185791015625f;  // 1 + 2^-22: boxes cull conservatively

constexpr uint32_t kInvalidPixel = 0xffffffffu;  // queue slot that holds no path (frame edge / foreign stripe)
// streaming accesses to the queues: every record is written once and read once or twice, nothing of it should displace the tree
typedef float v4f __attribute__((ext_vector_type(4)));
PTR_DEV float4 stream_load(const float4* p) {
    const v4f r = __builtin_nontemporal_load((const v4f*)p);
    return make_float4(r.x, r.y, r.z, r.w);
}
PTR_DEV void stream_store(float4* p, float4 v) {
    const v4f r = { v.x, v.y, v.z, v.w };
    __builtin_nontemporal_store(r, (v4f*)p);
}


// ---- generate ------------------------------------------------------------------------------------------
// One wave per 8x8 tile of the rows this GPU owns; slot = wave * 64 + lane, so the primary-ray queue needs
// no atomics and a wave's 64 rays are one screen tile (coherent traversal).  Lanes outside the frame keep
// their slot but mark it invalid.
__global__ __launch_bounds__(256) void ptr_wf_generate(SceneView sv, WavefrontScratch w, uint32_t domain_w, uint32_t domain_h,
                                                      uint32_t tiles_x, uint32_t owned_tile_rows) {
    const SceneMeta& m = sv.meta;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    // a dispatch of S samples queues every tile S times, the S copies next to each other: waves that run at the same time
    // walk the same part of the tree
    const uint32_t S = sv.batch_samples;
    if (wave >= tiles_x * owned_tile_rows * S) return;
    if (wave == 0u && lane == 0u) w.counters[0] = tiles_x * owned_tile_rows * S * 64u;
    const uint32_t tile = wave / S, sample_in_batch = wave % S;
    const uint32_t j = tile / tiles_x;  // j-th tile row owned by this GPU
    uint32_t tile_row = j;
    if (sv.part_world > 1u) {
        const uint32_t rows_per_stripe = sv.part_stripe_rows / 8u;
        tile_row = ((j / rows_per_stripe) * sv.part_world + sv.part_rank) * rows_per_stripe + j % rows_per_stripe;
    }
    const uint32_t gx = (tile % tiles_x) * 8u + (lane & 7u);
    const uint32_t gy = tile_row * 8u + (lane >> 3);
    const bool valid = gx < domain_w && gy < domain_h;

    Ray r; r.origin = mk3(0, 0, 0); r.dir = mk3(0, 0, 1);
    uint32_t seed = 0u;
    if (valid) {
        const int32_t cx = (int32_t)gx * m.downscale_factor, cy = (int32_t)gy * m.downscale_factor;
        seed = pixel_seed(cx, cy, seed_sample_index(sv, m.sample_index + sample_in_batch));
        r = generate_camera_ray(m, cx, cy, seed);  // seed by value: the bounce loop restarts from it
        if (sv.capture) { Capture c; c.prim0 = -1; c.t0 = 0.0f; c.prim1 = -1; c.t1 = 0.0f; sv.capture[(size_t)gy * m.width + gx] = c; }
    }
    const uint32_t slot = wave * 64u + lane;
    // a path names where its sample goes: sample k of the batch has its own width * height block of w.samples
    w.ray_o[0][slot] = make_float4(r.origin.x, r.origin.y, r.origin.z,
                                   __uint_as_float(valid ? sample_in_batch * (m.width * m.height) + gy * m.width + gx : kInvalidPixel));
    w.ray_d[0][slot] = make_float4(r.dir.x, r.dir.y, r.dir.z, __uint_as_float(seed));
    // (the throughput record of a primary ray is not written: it is 1, and the shade step of bounce 0 knows that)
}

// ---- extend --------------------------------------------------------------------------------------------
constexpr int32_t kEmptyChild = INT32_MIN;

// Entry distances of the four child boxes of a 4-wide node, +inf for a child the ray misses.
// Per node and axis the caller folds the ray into A = step / d and B = (origin - o) / d, so the distance to the
// plane at quantised offset q is ONE fused multiply-add, t = fma(q, A, B); the sign of 1/d has already picked
// which dword holds the entry planes and which the exit planes (one byte per child).  The 24 plane distances are
// computed two children at a time with the packed fp32 FMA (same IEEE result per element, half the instructions).
// v_max3 / v_min3 ignore NaN distances (0 * inf).  The host pads the quantised boxes by 2^-20 of the node extent,
// which covers the rounding of A, B and the fma; the (1 + 2^-22) factor covers the part of the error that is
// relative to t.
typedef float v2f __attribute__((ext_vector_type(2)));
PTR_DEV v2f plane_pair(uint32_t lo_byte, uint32_t hi_byte, float a, float b) {
    const v2f r = { __builtin_fmaf((float)lo_byte, a, b), __builtin_fmaf((float)hi_byte, a, b) };
    return r;
}
PTR_DEV float entry_key(float tnx, float tny, float tnz, float tfx, float tfy, float tfz, float min_t, float best_t) {
    const float t0 = max_num(max_num(max_num(tnx, tny), tnz), min_t);
    const float t1 = min_num(min_num(min_num(tfx, tfy), tfz), best_t);
    return (t1 * kBoxPad >= t0) ? t0 : __builtin_inff();
}
PTR_DEV void child_entries(uint32_t near_x, uint32_t near_y, uint32_t near_z, uint32_t far_x, uint32_t far_y, uint32_t far_z, const f3 A, const f3 B,
                           float min_t, float best_t, float& k0, float& k1, float& k2, float& k3) {
    const v2f nx01 = plane_pair(near_x & 0xffu, (near_x >> 8) & 0xffu, A.x, B.x), nx23 = plane_pair((near_x >> 16) & 0xffu, near_x >> 24, A.x, B.x);
    const v2f ny01 = plane_pair(near_y & 0xffu, (near_y >> 8) & 0xffu, A.y, B.y), ny23 = plane_pair((near_y >> 16) & 0xffu, near_y >> 24, A.y, B.y);
    const v2f nz01 = plane_pair(near_z & 0xffu, (near_z >> 8) & 0xffu, A.z, B.z), nz23 = plane_pair((near_z >> 16) & 0xffu, near_z >> 24, A.z, B.z);
    const v2f fx01 = plane_pair(far_x & 0xffu, (far_x >> 8) & 0xffu, A.x, B.x), fx23 = plane_pair((far_x >> 16) & 0xffu, far_x >> 24, A.x, B.x);
    const v2f fy01 = plane_pair(far_y & 0xffu, (far_y >> 8) & 0xffu, A.y, B.y), fy23 = plane_pair((far_y >> 16) & 0xffu, far_y >> 24, A.y, B.y);
    const v2f fz01 = plane_pair(far_z & 0xffu, (far_z >> 8) & 0xffu, A.z, B.z), fz23 = plane_pair((far_z >> 16) & 0xffu, far_z >> 24, A.z, B.z);
    k0 = entry_key(nx01.x, ny01.x, nz01.x, fx01.x, fy01.x, fz01.x, min_t, best_t);
    k1 = entry_key(nx01.y, ny01.y, nz01.y, fx01.y, fy01.y, fz01.y, min_t, best_t);
    k2 = entry_key(nx23.x, ny23.x, nz23.x, fx23.x, fy23.x, fz23.x, min_t, best_t);
    k3 = entry_key(nx23.y, ny23.y, nz23.y, fx23.y, fy23.y, fz23.y, min_t, best_t);
}

// compare-exchange of (key, reference) pairs, select form (no divergent branch)
PTR_DEV void sort_pair(float& ka, int32_t& ra, float& kb, int32_t& rb) {
    const bool swap = ka > kb;
    const float k_lo = swap ? kb : ka, k_hi = swap ? ka : kb;
    const int32_t r_lo = swap ? rb : ra, r_hi = swap ? ra : rb;
    ka = k_lo; kb = k_hi; ra = r_lo; rb = r_hi;
}

// Pop of the per-lane traversal stack: entry `sp` lives in LDS when sp < kStackLds (the common case: one
// ds_read), deeper entries in the global spill area (rare, separate branch so that the compiler keeps the two
// address spaces apart instead of falling back to a flat load).
PTR_DEV int32_t stack_pop(const int32_t* lds, const int32_t* spill, uint32_t sp) {
    int32_t v = lds[(sp < kStackLds ? sp : 0u) * 64u];
    // keep the LDS read a ds_read: without this fence the compiler folds both reads into one flat load of a selected
    // address, which goes through the vector-memory pipeline on every pop
    asm volatile("" : "+v"(v));
    if (sp >= kStackLds) v = spill[(sp - kStackLds) * 64u];
    return v;
}

// wave-uniform copies of lane L's / the first lane's value: they live in scalar registers
PTR_DEV float lane_value(float v, int lane_index) { return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), lane_index)); }
PTR_DEV int32_t lane_value(int32_t v, int lane_index) { return __builtin_amdgcn_readlane(v, lane_index); }
PTR_DEV uint32_t lane_value(uint32_t v, int lane_index) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane_index); }
PTR_DEV float uniform_value(float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v))); }
PTR_DEV int32_t uniform_value(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// A record of the layout's 64-byte primitive array that holds a sphere instead of a triangle (third float4's w = 1): centre in
// the first float4, radius in the second one's x.  The test is the reference's hit_sphere (ray.h:105-135) against the ray
// shortened to the best hit so far, which is how its commented call site in hit_node would run it (ray.h:173-177); a NaN
// distance is no hit, and the barycentrics a triangle hit carries are zero.
constexpr int32_t kRecordSphere = 1;
PTR_DEV void sphere_leaf(float4 a, float4 b, f3 o, f3 d, float min_t, float best_t, float& t, float& u, float& v, bool& reject) {
    Sphere s;
    s.position = make_float4(a.x, a.y, a.z, 0.0f);
    s.radius = b.x;
    Ray r; r.origin = o; r.dir = d; r.min_t = min_t; r.max_t = best_t;
    float ts = 0.0f;
    const bool hit = hit_sphere(s, r, ts);
    t = ts; u = 0.0f; v = 0.0f;
    reject = !hit || ts != ts;
}

constexpr int32_t kTraversalDone = INT32_MIN;  // never a leaf reference: triangle counts stay below 2^29

// Persistent wave64 traversal.
//   * Work assignment: wave g of the G resident waves owns the 64-ray batches g, g+G, g+2G, ... of the queue
//     and walks them with a private cursor -- no atomics (one address sustains only ~90 atomics/us), and the
//     batches of a wave are spread over the whole frame, which evens out the load.
//   * Ballot-driven refill: whenever at least REFILL lanes are idle (their ray finished) and the wave still owns
//     unread rays, the idle lanes fetch the next rays of the wave's list; a wave therefore keeps most lanes
//     busy instead of draining a batch down to its slowest ray.  (The memory pipeline's cost is per wave
//     instruction, not per active lane, so lane utilisation is what buys throughput.)
//   * Majority scheduling: a lane is either at an inner node or at a leaf.  Each wave iteration executes ONE of
//     the two step kinds, the one more lanes are waiting for (leaves are favoured by the factor LEAF_WEIGHT: a
//     leaf hit shortens the ray and culls later nodes); the other lanes keep their state and wait.  A ray's own
//     sequence of steps never changes, only when it is executed.
//   * The grid is the same for every launch of a frame.  Sharing a short queue among fewer waves (so that each keeps
//     ~300 rays and drains less) was measured: slower at every setting -- parallelism beats lane utilisation here.
//   * Per-lane stack: kStackLds entries in LDS (lane-interleaved, conflict-free), deeper entries in global memory.
//   * TWO_LEVEL: the tree in fv.top is walked first; its leaves name placed copies of the tree in fv.  A lane that
//     reaches one takes its ray into the object's space, stacks a marker and goes on at the object's root; when the
//     marker pops, the world-space ray is read back from the queue.  Both trees share the lane's one stack.
//     Entering a copy costs the lane a leaf turn (the record's loads, the transform, three reciprocals); doing it where the
//     reference becomes current, or as one repetition of an inner turn, was measured: 10 % slower -- the record loads stall
//     the whole wave inside its inner turn, and the extra live registers cost a resident wave.  Leaving costs nothing of
//     the kind: the world-space ray and its reciprocals wait in nine words of LDS per lane and come back where the marker
//     is popped (re-reading the queue there, and taking a leaf turn for it, made the placed copies 37 % slower than the
//     same copies written out).
//   * ANY_HIT: the first hit a lane finds ends its ray (the stack is dropped); used for the last bounce of a path, whose shade
//     step only asks whether the ray escaped.
template <bool STATS, int REFILL, int LEAF_WEIGHT, bool LEAF_ALL, int INNER_REPS, bool TWO_LEVEL, bool ANY_HIT = false>
__global__ __launch_bounds__(kExtendThreads, TWO_LEVEL ? 6 : 4) void ptr_wf_extend(FlatView fv, WavefrontScratch w, uint32_t gen, uint32_t slot_b,
                                                                unsigned long long* stats) {
    constexpr uint32_t kWaveWords = kStackLds * 64u + (TWO_LEVEL ? 9u * 64u : 0u);   // a wave's stacks, then its lanes' world-space rays
    __shared__ int32_t lds_stack[(kExtendThreads / 64) * kWaveWords];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_in_block = threadIdx.x >> 6;
    int32_t* my_lds = lds_stack + wave_in_block * kWaveWords + lane;
    float* my_world = (float*)(my_lds + kStackLds * 64u);   // TWO_LEVEL: o, d, 1/d of the ray outside the copy the lane is in
    int32_t* my_spill = w.stack_spill + (size_t)(blockIdx.x * (kExtendThreads / 64) + wave_in_block) * w.spill_entries * 64 + lane;

    const uint32_t count = w.counters[slot_b];
    const uint32_t global_wave = blockIdx.x * (kExtendThreads / 64) + wave_in_block;
    const uint32_t total_waves = gridDim.x * (kExtendThreads / 64);
    const uint32_t batches = (count + 63u) / 64u;
    const uint32_t my_batches = batches > global_wave ? (batches - global_wave + total_waves - 1u) / total_waves : 0u;
    const uint32_t my_rays = my_batches * 64u;  // entries past `count` or marked invalid are skipped when read
    uint32_t next_k = 0u;                       // wave-uniform: rays of my list already handed to lanes
    unsigned long long n_rays = 0, n_nodes = 0, n_tris = 0;
    unsigned long long wave_inner_steps = 0, wave_leaf_steps = 0;  // wave-uniform: iterations of each kind
    unsigned long long idle_lane_steps = 0;                        // wave-uniform: lanes without a ray, summed over inner-step slots

    const float min_t = 0.001f;
    uint32_t slot = kInvalidPixel;              // queue slot this lane is tracing; kInvalidPixel = idle
    f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), inv = mk3(0, 0, 1);
    bool neg_x = false, neg_y = false, neg_z = false;  // sign bits of 1/d: which plane of a slab is the entry
    float best_t = 1e15f, best_u = 0.0f, best_v = 0.0f;
    int32_t best_rank = -1, best_prim = -1;
    int32_t cur_inst = -1, best_inst = -1;       // TWO_LEVEL: the copy the lane is inside (-1: top level), the copy of the best hit
    int32_t cur = kTraversalDone;
    uint32_t sp = 0u;
    uint32_t alone_turns = 0u;                  // wave-uniform: iterations with a single busy lane and nothing left to fetch
    int lone_lane = -1;
    constexpr int32_t kLeaveMarker = ~(int32_t)(kInstanceTag | kLeaveInstance);
    // The next pending reference becomes current.  A marker on the way means the copy the lane was in is done: the lane is
    // back at the top level with the ray it had there.
    auto next_pending = [&]() {
        if (sp == 0u) { cur = kTraversalDone; return; }
        sp--;
        cur = stack_pop(my_lds, my_spill, sp);
        if (TWO_LEVEL && cur == kLeaveMarker) {
            o = mk3(my_world[0], my_world[64], my_world[128]);
            d = mk3(my_world[192], my_world[256], my_world[320]);
            inv = mk3(my_world[384], my_world[448], my_world[512]);
            neg_x = (as_u32(inv.x) >> 31) != 0u; neg_y = (as_u32(inv.y) >> 31) != 0u; neg_z = (as_u32(inv.z) >> 31) != 0u;
            cur_inst = -1;
            if (sp == 0u) { cur = kTraversalDone; return; }
            sp--;
            cur = stack_pop(my_lds, my_spill, sp);
        }
    };

    for (;;) {
        const unsigned long long idle = __ballot(slot == kInvalidPixel);
        if (idle != 0ull && next_k < my_rays && ((uint32_t)__popcll(idle) >= (uint32_t)REFILL || idle == ~0ull)) {
            if (slot == kInvalidPixel) {
                const uint32_t k = next_k + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                if (k < my_rays) {
                    const uint32_t idx = ((k >> 6) * total_waves + global_wave) * 64u + (k & 63u);
                    float4 ro = make_float4(0, 0, 0, __uint_as_float(kInvalidPixel));
                    if (idx < count) ro = w.ray_o[gen][idx];
                    if (__float_as_uint(ro.w) != kInvalidPixel) {
                        const float4 rd = w.ray_d[gen][idx];
                        slot = idx;
                        o = mk3(ro.x, ro.y, ro.z);
                        d = mk3(rd.x, rd.y, rd.z);
                        inv = mk3(rcp(d.x), rcp(d.y), rcp(d.z));
                        neg_x = (as_u32(inv.x) >> 31) != 0u; neg_y = (as_u32(inv.y) >> 31) != 0u; neg_z = (as_u32(inv.z) >> 31) != 0u;
                        best_t = 1e15f; best_u = 0.0f; best_v = 0.0f; best_rank = -1; best_prim = -1;
                        cur = TWO_LEVEL ? fv.top.root : fv.root; sp = 0u;
                        cur_inst = -1; best_inst = -1;
                        if (STATS) n_rays++;
                    }
                }
            }
            next_k += (uint32_t)__popcll(idle);
            continue;  // re-evaluate: skipped entries leave lanes idle
        }
        if (idle == ~0ull) break;  // nothing in flight and my list is exhausted
        if (!STATS && next_k >= my_rays && __popcll(~idle) == 1) {
            // one ray left in the wave and nothing to fetch: after a while the whole wave finishes it together (below)
            if (++alone_turns > w.alone_turns) { lone_lane = (int)__ffsll((long long)~idle) - 1; break; }
        }

        const bool at_inner = slot != kInvalidPixel && cur >= 0;
        const bool at_leaf = slot != kInvalidPixel && cur < 0;
        const uint32_t n_inner = (uint32_t)__popcll(__ballot(at_inner));
        const uint32_t n_leaf = (uint32_t)__popcll(__ballot(at_leaf));
        const bool leaf_turn = n_leaf * (uint32_t)LEAF_WEIGHT > n_inner;
        if (STATS) {
            if (leaf_turn) wave_leaf_steps++;
            else { wave_inner_steps += INNER_REPS; idle_lane_steps += (unsigned long long)INNER_REPS * (unsigned)__popcll(idle); }
        }

        // kInnerReps 4-wide nodes per inner turn (four box tests each, nearest child next, the others stacked):
        // repeating the step amortises the per-iteration bookkeeping (ballots, refill test, completion)
        #pragma unroll 1
        for (int rep = 0; rep < INNER_REPS; rep++)
        if (!leaf_turn && slot != kInvalidPixel && cur >= 0) {
            const float4* level_nodes = (TWO_LEVEL && cur_inst < 0) ? fv.top.nodes : fv.nodes;
            const float4 q0 = level_nodes[4 * cur];                      // origin.xyz, step.x
            const int4 refs = ((const int4*)level_nodes)[4 * cur + 1];
            const uint4 qa = ((const uint4*)level_nodes)[4 * cur + 2];  // lo.x lo.y lo.z hi.x
            const uint4 qb = ((const uint4*)level_nodes)[4 * cur + 3];  // hi.y hi.z step.y step.z
            if (STATS) n_nodes++;
            const f3 A = mk3(q0.w * inv.x, __uint_as_float(qb.z) * inv.y, __uint_as_float(qb.w) * inv.z);
            const f3 B = mk3((q0.x - o.x) * inv.x, (q0.y - o.y) * inv.y, (q0.z - o.z) * inv.z);
            const uint32_t near_x = neg_x ? qa.w : qa.x, far_x = neg_x ? qa.x : qa.w;
            const uint32_t near_y = neg_y ? qb.x : qa.y, far_y = neg_y ? qa.y : qb.x;
            const uint32_t near_z = neg_z ? qb.y : qa.z, far_z = neg_z ? qa.z : qb.y;
            int32_t r0 = refs.x, r1 = refs.y, r2 = refs.z, r3 = refs.w;
            const float inf = __builtin_inff();
            float k0, k1, k2, k3;
            child_entries(near_x, near_y, near_z, far_x, far_y, far_z, A, B, min_t, best_t, k0, k1, k2, k3);
            if (r2 == kEmptyChild) k2 = inf;  // children fill the slots from the front; 0 and 1 always exist
            if (r3 == kEmptyChild) k3 = inf;
            // Three compare-exchanges bring the nearest entry to slot 0 and the farther of the two pair winners -- most
            // often the second nearest -- to slot 2; slots 1 and 3 keep the loser of each pair.  The full five-exchange sort
            // costs ten more instructions per node and saves under 1 % of the node visits (the visiting order decides
            // only how early a hit shortens the ray, never which hit is found).
            sort_pair(k0, r0, k1, r1);
            sort_pair(k2, r2, k3, r3);
            sort_pair(k0, r0, k2, r2);
            if (k0 < inf) {
                // the other children the ray enters go on the stack: slot 3, slot 1, then slot 2 on top
                if (sp + 3u <= kStackLds) {
                    // common case, branch-free: every reference is written at the current top and the top advances only
                    // for a hit, so exactly the hit ones end up below the new top (a miss is overwritten or left above it)
                    my_lds[sp * 64u] = r3; sp += (k3 < inf) ? 1u : 0u;
                    my_lds[sp * 64u] = r1; sp += (k1 < inf) ? 1u : 0u;
                    my_lds[sp * 64u] = r2; sp += (k2 < inf) ? 1u : 0u;
                } else {
                    if (k3 < inf) { if (sp < kStackLds) my_lds[sp * 64u] = r3; else my_spill[(sp - kStackLds) * 64u] = r3; sp++; }
                    if (k1 < inf) { if (sp < kStackLds) my_lds[sp * 64u] = r1; else my_spill[(sp - kStackLds) * 64u] = r1; sp++; }
                    if (k2 < inf) { if (sp < kStackLds) my_lds[sp * 64u] = r2; else my_spill[(sp - kStackLds) * 64u] = r2; sp++; }
                }
                cur = r0;
            } else {
                next_pending();
            }
        }

        if (leaf_turn && at_leaf) {
            uint32_t ref = (uint32_t)~cur;
            if (TWO_LEVEL && (ref & kInstanceTag) != 0u) {
                // a placed copy: the ray goes into the object's space (the one it has out here is kept in LDS), a marker goes
                // on the stack, and the walk goes on at the object's root
                const uint32_t index = ref & ~kInstanceTag;
                const float4* rec = fv.top.records + 4 * (size_t)index;
                const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
                const int32_t object_root = __float_as_int(rec[3].x);
                const f3 wo = o, wd = d;
                my_world[0] = wo.x; my_world[64] = wo.y; my_world[128] = wo.z;
                my_world[192] = wd.x; my_world[256] = wd.y; my_world[320] = wd.z;
                my_world[384] = inv.x; my_world[448] = inv.y; my_world[512] = inv.z;
                o = xform_point(r0, r1, r2, wo);
                d = xform_dir(r0, r1, r2, wd);
                cur_inst = (int32_t)index;
                if (sp < kStackLds) my_lds[sp * 64u] = kLeaveMarker; else my_spill[(sp - kStackLds) * 64u] = kLeaveMarker;
                sp++;
                cur = object_root;
                // a copy that is only moved, not turned or scaled, leaves the direction as it is, bit for bit: its reciprocals
                // (three correctly rounded divisions) are the ones the lane already has
                if (as_u32(d.x) != as_u32(wd.x) || as_u32(d.y) != as_u32(wd.y) || as_u32(d.z) != as_u32(wd.z)) {
                    inv = mk3(rcp(d.x), rcp(d.y), rcp(d.z));
                    neg_x = (as_u32(inv.x) >> 31) != 0u; neg_y = (as_u32(inv.y) >> 31) != 0u; neg_z = (as_u32(inv.z) >> 31) != 0u;
                }
            } else {
            // LEAF_ALL: every triangle of the lane's leaf in this turn; otherwise ONE triangle, the rest of the leaf
            // stays encoded in `cur` for the next leaf turn (lanes with short leaves get back to inner nodes sooner)
            for (;;) {
                const uint32_t first = ref >> 2, remaining = ref & 3u;  // triangles left after this one
                const float4* tr = fv.triangles + 4 * (size_t)first;
                const float4 a = tr[0], b = tr[1], c = tr[2], n4 = tr[3];
                if (STATS) n_tris++;
                const f3 e1 = mk3(b.x, b.y, b.z), e2 = mk3(c.x, c.y, c.z), n = mk3(n4.x, n4.y, n4.z);
                const f3 ov = o - mk3(a.x, a.y, a.z);
                const f3 q = cross(ov, d);
                const float dd = rcp(dot(d, n));
                float u = dd * dot(-q, e2);
                float v = dd * dot(q, e1);
                float t = dd * dot(-n, ov);
                bool reject = u < 0.0f || u > 1.0f || v < 0.0f || u + v > 1.0f || t < min_t || t > best_t || t != t || u != u || v != v;
                if (!TWO_LEVEL && __float_as_int(c.w) == kRecordSphere) sphere_leaf(a, b, o, d, min_t, best_t, t, u, v, reject);
                const int32_t rank = __float_as_int(a.w);
                // equal t: the later (copy, rank) pair wins
                const bool earlier = TWO_LEVEL ? (cur_inst < best_inst || (cur_inst == best_inst && rank < best_rank)) : rank < best_rank;
                if (!reject && !(best_prim >= 0 && t == best_t && earlier)) {
                    best_t = t; best_u = u; best_v = v; best_rank = rank; best_prim = __float_as_int(b.w);
                    if (TWO_LEVEL) best_inst = cur_inst;
                    if (ANY_HIT) { ref = 0xffffffffu; sp = 0u; break; }   // nothing pending matters any more
                }
                if (remaining == 0u) { ref = 0xffffffffu; break; }
                ref = ((first + 1u) << 2) | (remaining - 1u);
                if (!LEAF_ALL) break;
            }
            if (ref != 0xffffffffu) cur = ~(int32_t)ref; else next_pending();
            }
        }

        // finished rays publish their hit and free the lane
        if (slot != kInvalidPixel && cur == kTraversalDone) {
            stream_store(&w.hits[slot], make_float4(best_t, best_u, best_v, __int_as_float(best_prim)));
            if (TWO_LEVEL) w.hit_instance[slot] = best_inst;
            slot = kInvalidPixel;
        }
    }

    // ---- the wave's last ray, finished by all 64 lanes -------------------------------------------------------------
    // A lone lane pays a full memory latency per node; rays that graze a long run of boxes (thousands of nodes) would
    // hold the whole launch for milliseconds.  Its pending references (stack + current) become one list in the wave's
    // LDS; every turn up to 64 of them are popped, one per lane: a node lane tests its four children and appends the
    // ones the ray enters, a leaf lane tests its triangles.  Candidates are reduced with the rule of the sequential
    // walk (smaller t, on equal t the later rank), which does not depend on the visiting order, so the hit is the same.
    // With placed copies every list entry carries the copy it belongs to (-1: top level) in a second array, the lane
    // takes the world-space ray into that copy's space itself, and an instance leaf appends the object's root under
    // the copy's number; the marker the single-lane walk keeps on its stack is not needed.
    if (!STATS && lone_lane >= 0) {
        int32_t* list = lds_stack + wave_in_block * kWaveWords;       // references, linear
        constexpr uint32_t kListMax = TWO_LEVEL ? kStackLds * 32 : kStackLds * 64;
        int32_t* list_copy = list + kListMax;                         // TWO_LEVEL: the copy of each entry
        // everything about the ray is the same in all lanes from here on: kept in scalar registers
        const int L = __builtin_amdgcn_readfirstlane(lone_lane);
        const uint32_t the_slot = lane_value(slot, L);
        f3 wo = mk3(lane_value(o.x, L), lane_value(o.y, L), lane_value(o.z, L));
        f3 wd = mk3(lane_value(d.x, L), lane_value(d.y, L), lane_value(d.z, L));
        if (TWO_LEVEL) {  // lane L may be inside a copy: the world-space ray is in the queue
            const float4 qo = w.ray_o[gen][the_slot], qd = w.ray_d[gen][the_slot];
            wo = mk3(uniform_value(qo.x), uniform_value(qo.y), uniform_value(qo.z));
            wd = mk3(uniform_value(qd.x), uniform_value(qd.y), uniform_value(qd.z));
        }
        float bt = lane_value(best_t, L), bu = lane_value(best_u, L), bv = lane_value(best_v, L);
        int32_t brank = lane_value(best_rank, L), bprim = lane_value(best_prim, L), bcopy = lane_value(best_inst, L);
        const int32_t copy_of_l = lane_value(cur_inst, L);
        const uint32_t depth = lane_value(sp, L);
        int32_t top = lane_value(cur, L);
        constexpr int32_t kMarker = kLeaveMarker;
        // gather the lane-interleaved stack of lane L, then lay it out linearly (at most kStackMax + 3 entries: four per lane)
        const int32_t* spill_of_l = my_spill - lane + L;
        int32_t held[4];
        uint32_t marker_at = 0xffffffffu;
        #pragma unroll
        for (uint32_t j = 0; j < 4u; j++) {
            const uint32_t e = lane + 64u * j;
            held[j] = kEmptyChild;
            if (e < depth) held[j] = e < kStackLds ? list[e * 64u + (uint32_t)L] : spill_of_l[(e - kStackLds) * 64u];
            if (TWO_LEVEL) {
                const unsigned long long found = __ballot(e < depth && held[j] == kMarker);
                if (found != 0ull) marker_at = (uint32_t)__ffsll((long long)found) - 1u + 64u * j;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        #pragma unroll
        for (uint32_t j = 0; j < 4u; j++) {
            const uint32_t e = lane + 64u * j;
            if (e < depth && e < kListMax) {
                list[e] = (TWO_LEVEL && e == marker_at) ? kEmptyChild : held[j];
                if (TWO_LEVEL) list_copy[e] = (marker_at != 0xffffffffu && e > marker_at) ? copy_of_l : -1;
            }
        }
        uint32_t size = depth < kListMax ? depth : kListMax;
        if (TWO_LEVEL && top == kMarker) top = kTraversalDone;   // the copy was finished: nothing of it is pending
        if (lane == 0u && top != kTraversalDone && size < kListMax) {
            list[size] = top;
            if (TWO_LEVEL) list_copy[size] = copy_of_l;
        }
        if (top != kTraversalDone) size++;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // The list is a ring in LDS, [tail, head); when a turn's appends might not fit, its oldest block moves out to the
        // wave's spill area in global memory (free: lane L's deep entries were copied out above), and comes back when the
        // ring runs low -- so a ray whose frontier is thousands of references wide still gets 64 lanes per turn.
        constexpr uint32_t kMask = kListMax - 1u;
        constexpr uint32_t kBlock = kListMax / 4u;
        int32_t* overflow = w.stack_spill + (size_t)global_wave * w.spill_entries * 64u;
        const uint32_t overflow_cap = ((w.spill_entries * 64u) / (TWO_LEVEL ? 2u : 1u)) / kBlock * kBlock;
        int32_t* overflow_copy = overflow + overflow_cap;              // TWO_LEVEL: the copies of the entries moved out
        // No pending reference is ever dropped.  A turn that pops p entries appends at most 4p, so the ring grows by at most
        // 3p.  Turns are sized so that the ring stays within `limit` = kListMax - (fv.max_depth + 3) entries; once it is that
        // full and the spill area is full too, turns pop ONE entry, the newest: the list then behaves as the stack of a
        // depth-first walk below that entry, which holds at most fv.max_depth + 1 references at once (the bound the layout
        // header states for a lane's stack, checked against the tree before the first dispatch; + 1 because the nearest child
        // is stacked here too), and the ring is back to its earlier length when that subtree is done.  So the ring never
        // exceeds limit + fv.max_depth + 1 < kListMax, whatever the width of the ray's frontier.
        const uint32_t limit = kListMax - (fv.max_depth + 3u);
        uint32_t tail = 0u, head = size, moved_out = 0u;
        while (head != tail || moved_out != 0u) {
            if (head - tail < 64u && moved_out != 0u) {                // bring the newest block back below the ring
                tail -= kBlock; moved_out -= kBlock;
                for (uint32_t i = lane; i < kBlock; i += 64u) {
                    list[(tail + i) & kMask] = overflow[moved_out + i];
                    if (TWO_LEVEL) list_copy[(tail + i) & kMask] = overflow_copy[moved_out + i];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            uint32_t pops = head - tail < 64u ? head - tail : 64u;
            if (head - tail + 3u * pops > limit && head - tail >= kBlock + 64u && moved_out + kBlock <= overflow_cap) {
                for (uint32_t i = lane; i < kBlock; i += 64u) {        // make room: the oldest block leaves the ring
                    overflow[moved_out + i] = list[(tail + i) & kMask];
                    if (TWO_LEVEL) overflow_copy[moved_out + i] = list_copy[(tail + i) & kMask];
                }
                tail += kBlock; moved_out += kBlock;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            if (head - tail + 3u * pops > limit)                        // still too full: a narrower turn, down to depth first
                pops = head - tail + 3u <= limit ? (limit - (head - tail)) / 3u : 1u;
            const bool mine = lane < pops;
            const int32_t ref = mine ? list[(head - 1u - lane) & kMask] : kEmptyChild;
            int32_t copy = -1;
            if (TWO_LEVEL && mine) copy = list_copy[(head - 1u - lane) & kMask];
            head -= pops;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();

            // this entry's ray: the world-space one, or that ray inside the entry's copy
            f3 ro = wo, rd = wd;
            if (TWO_LEVEL && copy >= 0) {
                const float4* rec = fv.top.records + 4 * (size_t)copy;
                const float4 m0 = rec[0], m1 = rec[1], m2 = rec[2];
                ro = xform_point(m0, m1, m2, wo);
                rd = xform_dir(m0, m1, m2, wd);
            }
            int32_t r0 = kEmptyChild, r1 = kEmptyChild, r2 = kEmptyChild, r3 = kEmptyChild;
            int32_t push_copy = copy;
            uint32_t n_push = 0u;
            float ct = __builtin_inff(), cu = 0.0f, cv = 0.0f;   // this lane's candidate hit
            int32_t crank = -1, cprim = -1, ccopy = -1;
            if (mine && ref >= 0) {
                const float4* level_nodes = (TWO_LEVEL && copy < 0) ? fv.top.nodes : fv.nodes;
                const f3 rinv = mk3(rcp(rd.x), rcp(rd.y), rcp(rd.z));
                const bool ngx = (as_u32(rinv.x) >> 31) != 0u, ngy = (as_u32(rinv.y) >> 31) != 0u, ngz = (as_u32(rinv.z) >> 31) != 0u;
                const float4 q0 = level_nodes[4 * ref];
                const int4 refs = ((const int4*)level_nodes)[4 * ref + 1];
                const uint4 qa = ((const uint4*)level_nodes)[4 * ref + 2];
                const uint4 qb = ((const uint4*)level_nodes)[4 * ref + 3];
                const f3 A = mk3(q0.w * rinv.x, __uint_as_float(qb.z) * rinv.y, __uint_as_float(qb.w) * rinv.z);
                const f3 B = mk3((q0.x - ro.x) * rinv.x, (q0.y - ro.y) * rinv.y, (q0.z - ro.z) * rinv.z);
                const uint32_t near_x = ngx ? qa.w : qa.x, far_x = ngx ? qa.x : qa.w;
                const uint32_t near_y = ngy ? qb.x : qa.y, far_y = ngy ? qa.y : qb.x;
                const uint32_t near_z = ngz ? qb.y : qa.z, far_z = ngz ? qa.z : qb.y;
                r0 = refs.x; r1 = refs.y; r2 = refs.z; r3 = refs.w;
                const float inf = __builtin_inff();
                float k0, k1, k2, k3;
                child_entries(near_x, near_y, near_z, far_x, far_y, far_z, A, B, min_t, bt, k0, k1, k2, k3);
                if (r2 == kEmptyChild) k2 = inf;
                if (r3 == kEmptyChild) k3 = inf;
                sort_pair(k0, r0, k1, r1);
                sort_pair(k2, r2, k3, r3);
                sort_pair(k0, r0, k2, r2);
                sort_pair(k1, r1, k3, r3);
                sort_pair(k1, r1, k2, r2);
                n_push = (k0 < inf ? 1u : 0u) + (k1 < inf ? 1u : 0u) + (k2 < inf ? 1u : 0u) + (k3 < inf ? 1u : 0u);  // hits are sorted to the front
            } else if (TWO_LEVEL && mine && ref != kEmptyChild && copy < 0) {
                // a leaf of the top level: the object's root, under the copy's number
                const uint32_t index = (uint32_t)~ref & ~kInstanceTag;
                r0 = __float_as_int(fv.top.records[4 * (size_t)index + 3].x);
                push_copy = (int32_t)index;
                n_push = 1u;
            } else if (mine && ref != kEmptyChild) {
                uint32_t lr = (uint32_t)~ref;
                for (;;) {
                    const uint32_t first = lr >> 2, remaining = lr & 3u;
                    const float4* tr = fv.triangles + 4 * (size_t)first;
                    const float4 a = tr[0], b = tr[1], c = tr[2], n4 = tr[3];
                    const f3 e1 = mk3(b.x, b.y, b.z), e2 = mk3(c.x, c.y, c.z), n = mk3(n4.x, n4.y, n4.z);
                    const f3 ov = ro - mk3(a.x, a.y, a.z);
                    const f3 q = cross(ov, rd);
                    const float dd = rcp(dot(rd, n));
                    float u = dd * dot(-q, e2);
                    float v = dd * dot(q, e1);
                    float t = dd * dot(-n, ov);
                    bool reject = u < 0.0f || u > 1.0f || v < 0.0f || u + v > 1.0f || t < min_t || t > bt || t != t || u != u || v != v;
                    if (!TWO_LEVEL && __float_as_int(c.w) == kRecordSphere) sphere_leaf(a, b, ro, rd, min_t, bt, t, u, v, reject);
                    const int32_t rank = __float_as_int(a.w);
                    if (!reject && (t < ct || (t == ct && rank > crank))) { ct = t; cu = u; cv = v; crank = rank; cprim = __float_as_int(b.w); ccopy = copy; }
                    if (remaining == 0u) break;
                    lr = ((first + 1u) << 2) | (remaining - 1u);
                }
            }
            // the best candidate of the turn: smallest t, then the later (copy, rank) pair
            if (__ballot(cprim >= 0) != 0ull) {
                #pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const float ot = __shfl_xor(ct, off), ou = __shfl_xor(cu, off), ov2 = __shfl_xor(cv, off);
                    const int32_t orank = __shfl_xor(crank, off), oprim = __shfl_xor(cprim, off);
                    const int32_t ocopy = TWO_LEVEL ? __shfl_xor(ccopy, off) : -1;
                    const bool later = TWO_LEVEL ? (ocopy > ccopy || (ocopy == ccopy && orank > crank)) : orank > crank;
                    const bool take = oprim >= 0 && (cprim < 0 || ot < ct || (ot == ct && later));
                    if (take) { ct = ot; cu = ou; cv = ov2; crank = orank; cprim = oprim; ccopy = ocopy; }
                }
                // after the butterfly every lane holds the same candidate
                ct = uniform_value(ct); cu = uniform_value(cu); cv = uniform_value(cv);
                crank = uniform_value(crank); cprim = uniform_value(cprim); ccopy = uniform_value(ccopy);
                const bool earlier = TWO_LEVEL ? (ccopy < bcopy || (ccopy == bcopy && crank < brank)) : crank < brank;
                if (cprim >= 0 && !(bprim >= 0 && ct == bt && earlier)) { bt = ct; bu = cu; bv = cv; brank = crank; bprim = cprim; bcopy = ccopy; }
            }
            if (ANY_HIT && bprim >= 0) break;   // wave-uniform
            // append the entered children: farthest first within a lane, so that a depth-first turn pops the nearest
            const unsigned long long b0 = __ballot((n_push & 1u) != 0u), b1 = __ballot((n_push & 2u) != 0u), b2 = __ballot((n_push & 4u) != 0u);
            const unsigned long long below = (1ull << lane) - 1ull;
            const uint32_t offset = (uint32_t)__popcll(b0 & below) + 2u * (uint32_t)__popcll(b1 & below) + 4u * (uint32_t)__popcll(b2 & below);
            const uint32_t total = (uint32_t)__popcll(b0) + 2u * (uint32_t)__popcll(b1) + 4u * (uint32_t)__popcll(b2);
            if (head - tail + total > kListMax) {
                // Unreachable for a layout whose header states its true stack need (see `limit` above); a caller who turned
                // the scene checks off can get here with a header that understates it.  Never write past the ring: report
                // through the host-visible fault word (the next ptr_sync / ptr_dispatch fails with PTR_E_RANGE) and stop.
                if (lane == 0u) *w.fault = 1u;
                break;
            }
            uint32_t at = head + offset;
            if (TWO_LEVEL) for (uint32_t j = 0; j < n_push; j++) list_copy[(at + j) & kMask] = push_copy;
            if (n_push > 3u) list[at++ & kMask] = r3;
            if (n_push > 2u) list[at++ & kMask] = r2;
            if (n_push > 1u) list[at++ & kMask] = r1;
            if (n_push > 0u) list[at++ & kMask] = r0;
            head += total;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0u) {
            w.hits[the_slot] = make_float4(bt, bu, bv, __int_as_float(bprim));
            if (TWO_LEVEL) w.hit_instance[the_slot] = bcopy;
        }
    }
    if (STATS && stats) {
        for (int off = 32; off > 0; off >>= 1) {
            n_rays += __shfl_down(n_rays, off);
            n_nodes += __shfl_down(n_nodes, off);
            n_tris += __shfl_down(n_tris, off);
        }
        if (lane == 0u) {
            atomicAdd(&stats[0], n_rays); atomicAdd(&stats[1], n_nodes); atomicAdd(&stats[2], n_tris);
            atomicAdd(&stats[5], wave_inner_steps); atomicAdd(&stats[6], wave_leaf_steps); atomicAdd(&stats[7], idle_lane_steps);
        }
    }
}

// ---- shade ---------------------------------------------------------------------------------------------
constexpr uint32_t kShadeThreads = 512;   // measured: 256 -> shade +5 % (twice the atomics on the queue counter)

// (Grouping the survivors by the octant of their new direction, so that a traversal wave holds rays that order a
// node's children alike, was measured: no gain in the traversal, a slower shade step.)
// Survivors of a block (kShadeThreads lanes) are compacted with ONE atomic on the next queue's counter: ballots give the
// per-wave counts, lane 0 of the block turns them into offsets.  (A single address sustains only ~90 atomics
// per microsecond, so per-wave appends would bound the kernel.)
template <bool STATS>
__global__ __launch_bounds__(kShadeThreads) void ptr_wf_shade(SceneView sv, WavefrontScratch w, uint32_t gen, uint32_t slot_b,
                                                             uint32_t slot_next, uint32_t bounce, uint32_t max_bounce, uint32_t min_bounce) {
    __shared__ uint32_t wave_offset[kShadeThreads / 64];
    __shared__ uint32_t block_base;
    const uint32_t count = w.counters[slot_b];
    if (blockIdx.x * kShadeThreads >= count) return;  // uniform for the block
    const uint32_t i = blockIdx.x * kShadeThreads + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float4 ro = make_float4(0, 0, 0, __uint_as_float(kInvalidPixel));
    if (i < count) ro = stream_load(&w.ray_o[gen][i]);
    const uint32_t pixel = __float_as_uint(ro.w);
    const bool live = pixel != kInvalidPixel;

    bool go_on = false;
    Ray r; r.origin = mk3(0, 0, 0); r.dir = mk3(0, 0, 0); r.min_t = 0.001f; r.max_t = 1e15f;
    f3 throughput = mk3(0, 0, 0), result = mk3(0, 0, 0);
    uint32_t seed = 0u;
    unsigned long long shaded = 0;
    if (live) {
        const float4 rd = stream_load(&w.ray_d[gen][i]);
        float4 pt = make_float4(1.0f, 1.0f, 1.0f, 0.0f);   // primary rays carry no throughput record
        if (bounce != 0u) pt = stream_load(&w.path_t[gen][i]);
        r.origin = mk3(ro.x, ro.y, ro.z);
        r.dir = mk3(rd.x, rd.y, rd.z);
        throughput = mk3(pt.x, pt.y, pt.z);
        seed = __float_as_uint(rd.w);
        const float4 h4 = stream_load(&w.hits[i]);
        Hit h; h.t = h4.x; h.u = h4.y; h.v = h4.z; h.prim = __float_as_int(h4.w);
        if (sv.instance_records && h.prim >= 0) h.inst = w.hit_instance[i];
        if (STATS && h.prim >= 0) shaded = 1;
        if (sv.capture && bounce < 2u) {
            Capture* c = &sv.capture[pixel];
            const int32_t named = h.prim >= 0 && h.inst >= 0 ? (int32_t)((uint32_t)h.inst * sv.object_triangles + (uint32_t)h.prim) : h.prim;
            if (bounce == 0u) { c->prim0 = named; c->t0 = h.prim >= 0 ? h.t : 0.0f; }
            else { c->prim1 = named; c->t1 = h.prim >= 0 ? h.t : 0.0f; }
        }
        if (!STATS && bounce + 1u >= max_bounce) {
            // the last turn of the bounce loop: a ray that escapes returns the path's throughput (compute.comp:130-136), and whatever
            // else happens to the path -- absorbed, ended by the roulette, or still alive when the loop runs out (compute.comp:241-244)
            // -- it returns black: only hit-or-miss is looked at (the traversal of this bounce stops at the first hit it finds)
            result = h.prim < 0 ? throughput * 1.0f : mk3(0.0f, 0.0f, 0.0f);
        } else {
            go_on = shade_bounce(sv, bounce, min_bounce, h, r, throughput, seed, result);
            if (go_on && bounce + 1u >= max_bounce) { go_on = false; result = mk3(0.0f, 0.0f, 0.0f); }  // loop exhausted: radiance
        }
        if (!go_on) stream_store(&w.samples[pixel], make_float4(result.x, result.y, result.z, 1.0f));
    }

    const unsigned long long mask = __ballot(go_on);
    if (lane == 0u) wave_offset[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0u) {
        uint32_t total = 0u;
        for (uint32_t k = 0; k < kShadeThreads / 64; k++) { const uint32_t n = wave_offset[k]; wave_offset[k] = total; total += n; }
        block_base = total ? atomicAdd(&w.counters[slot_next], total) : 0u;
    }
    __syncthreads();
    if (go_on) {
        const uint32_t next = gen ^ 1u;
        const uint32_t slot = block_base + wave_offset[wave] + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        stream_store(&w.ray_o[next][slot], make_float4(r.origin.x, r.origin.y, r.origin.z, __uint_as_float(pixel)));
        stream_store(&w.ray_d[next][slot], make_float4(r.dir.x, r.dir.y, r.dir.z, __uint_as_float(seed)));
        stream_store(&w.path_t[next][slot], make_float4(throughput.x, throughput.y, throughput.z, 0.0f));
    }
    if (STATS && sv.stats) {
        for (int off = 32; off > 0; off >>= 1) shaded += __shfl_down(shaded, off);
        if (lane == 0u && shaded) atomicAdd(&sv.stats[3], shaded);
    }
}

// ---- accumulate ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ptr_wf_accumulate(SceneView sv, WavefrontScratch w, uint32_t domain_w, uint32_t domain_h,
                                                        uint32_t tiles_x) {
    const SceneMeta& m = sv.meta;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gx = (wave % tiles_x) * 8u + (lane & 7u);
    const uint32_t gy = (wave / tiles_x) * 8u + (lane >> 3);
    if (gx >= domain_w || gy >= domain_h || !owns_row(sv, gy)) return;
    if (sv.batch_samples > 1u) {
        accumulate_pixel_chain(sv.images[sv.accumulation_index], sv.images[sv.output_index],
                               sv.prev_image_plus_one ? &sv.images[sv.prev_image_plus_one - 1u] : nullptr, (int32_t)gx * m.downscale_factor,
                               (int32_t)gy * m.downscale_factor, m.downscale_factor, m.sample_index, sv.batch_samples,
                               w.samples + (gy * m.width + gx), (size_t)m.width * m.height);
        return;
    }
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
    const uint32_t tiles_x = (domain_w + 7u) / 8u;
    uint32_t owned_tile_rows = (domain_h + 7u) / 8u;
    if (sv.part_world > 1u) {
        const uint32_t rows_per_stripe = sv.part_stripe_rows / 8u;
        const uint32_t stripes = (owned_tile_rows + rows_per_stripe - 1u) / rows_per_stripe;
        const uint32_t owned = stripes > sv.part_rank ? (stripes - sv.part_rank + sv.part_world - 1u) / sv.part_world : 0u;
        owned_tile_rows = owned * rows_per_stripe;
    }
    const uint32_t waves = tiles_x * owned_tile_rows * sv.batch_samples;
    // the kernel publishes the queue length; with nothing owned, leave the zeroed counter alone
    if (waves == 0) return hipSuccess;
    hipLaunchKernelGGL(ptr_wf_generate, dim3((waves + 3u) / 4u), dim3(256), 0, s, sv, w, domain_w, domain_h, tiles_x, owned_tile_rows);
    return hipGetLastError();
}

template <bool STATS, int LEAF_WEIGHT, bool LEAF_ALL, int INNER_REPS, bool TWO_LEVEL = false, bool ANY_HIT = false>
static void launch_extend_variant(hipStream_t s, const FlatView& fv, const WavefrontScratch& w, uint32_t bounce, unsigned long long* stats,
                                  int refill) {
    const uint32_t slot_b = bounce % kMaxBounceSlots;
    const dim3 grid(w.extend_blocks), block(kExtendThreads);
    switch (refill) {
    case 8: hipLaunchKernelGGL((ptr_wf_extend<STATS, 8, LEAF_WEIGHT, LEAF_ALL, INNER_REPS, TWO_LEVEL, ANY_HIT>), grid, block, 0, s, fv, w, bounce & 1u, slot_b, stats); break;
    case 16: hipLaunchKernelGGL((ptr_wf_extend<STATS, 16, LEAF_WEIGHT, LEAF_ALL, INNER_REPS, TWO_LEVEL, ANY_HIT>), grid, block, 0, s, fv, w, bounce & 1u, slot_b, stats); break;
    case 32: hipLaunchKernelGGL((ptr_wf_extend<STATS, 32, LEAF_WEIGHT, LEAF_ALL, INNER_REPS, TWO_LEVEL, ANY_HIT>), grid, block, 0, s, fv, w, bounce & 1u, slot_b, stats); break;
    default: hipLaunchKernelGGL((ptr_wf_extend<STATS, 64, LEAF_WEIGHT, LEAF_ALL, INNER_REPS, TWO_LEVEL, ANY_HIT>), grid, block, 0, s, fv, w, bounce & 1u, slot_b, stats); break;
    }
}

hipError_t launch_wf_extend(hipStream_t s, const FlatView& fv, const WavefrontScratch& w, uint32_t bounce, unsigned long long* stats,
                            int refill, int leaf_weight, int leaf_all, int inner_reps, bool any_hit) {
    // (the counting variants and the tuning configurations always look for the closest hit: the counts they report are the
    // oracle's, and a closest hit is an answer to "any hit?" as well)
    if (fv.top.nodes) {  // placed copies: one configuration, the default one
        if (stats) launch_extend_variant<true, 2, true, 3, true>(s, fv, w, bounce, stats, refill);
        else if (any_hit) launch_extend_variant<false, 2, true, 3, true, true>(s, fv, w, bounce, stats, refill);
        else launch_extend_variant<false, 2, true, 3, true>(s, fv, w, bounce, stats, refill);
    } else if (!stats && any_hit && leaf_all && inner_reps >= 3) {
        launch_extend_variant<false, 2, true, 3, false, true>(s, fv, w, bounce, stats, refill);
    } else if (stats) {  // the counting variant follows inner_reps so that its step counts describe the timed configuration
        if (inner_reps >= 3) launch_extend_variant<true, 2, true, 3>(s, fv, w, bounce, stats, refill);
        else launch_extend_variant<true, 2, true, 1>(s, fv, w, bounce, stats, refill);
    } else if (!leaf_all) {
        launch_extend_variant<false, 2, false, 1>(s, fv, w, bounce, stats, refill);
    } else if (inner_reps == 1) {
        switch (leaf_weight) {
        case 1: launch_extend_variant<false, 1, true, 1>(s, fv, w, bounce, stats, refill); break;
        case 4: launch_extend_variant<false, 4, true, 1>(s, fv, w, bounce, stats, refill); break;
        default: launch_extend_variant<false, 2, true, 1>(s, fv, w, bounce, stats, refill); break;
        }
    } else if (inner_reps == 2) {
        switch (leaf_weight) {
        case 1: launch_extend_variant<false, 1, true, 2>(s, fv, w, bounce, stats, refill); break;
        case 4: launch_extend_variant<false, 4, true, 2>(s, fv, w, bounce, stats, refill); break;
        default: launch_extend_variant<false, 2, true, 2>(s, fv, w, bounce, stats, refill); break;
        }
    } else {
        launch_extend_variant<false, 2, true, 3>(s, fv, w, bounce, stats, refill);
    }
    return hipGetLastError();
}

hipError_t launch_wf_shade(hipStream_t s, const SceneView& sv, const WavefrontScratch& w, uint32_t bounce, uint32_t max_bounce,
                           uint32_t min_bounce, uint32_t domain_w, uint32_t domain_h) {
    // queue slots never exceed the padded tile count of the frame
    const uint32_t upper = ((domain_w + 7u) / 8u) * ((domain_h + 7u) / 8u + sv.part_stripe_rows / 8u) * 64u * sv.batch_samples;
    const uint32_t blocks = (upper + kShadeThreads - 1u) / kShadeThreads;
    if (blocks == 0) return hipSuccess;
    const uint32_t slot_b = bounce % kMaxBounceSlots, slot_next = (bounce + 1u) % kMaxBounceSlots;
    if (sv.stats) hipLaunchKernelGGL(ptr_wf_shade<true>, dim3(blocks), dim3(kShadeThreads), 0, s, sv, w, bounce & 1u, slot_b, slot_next, bounce, max_bounce, min_bounce);
    else hipLaunchKernelGGL(ptr_wf_shade<false>, dim3(blocks), dim3(kShadeThreads), 0, s, sv, w, bounce & 1u, slot_b, slot_next, bounce, max_bounce, min_bounce);
    return hipGetLastError();
}

// One lane per triangle: the shade step's inputs gathered once into a contiguous record.
__global__ __launch_bounds__(256) void ptr_wf_build_shade_records(SceneView sv, uint32_t triangle_count, uint32_t vertex_count, float4* records) {
    const uint32_t tri = blockIdx.x * 256u + threadIdx.x;
    if (tri >= triangle_count) return;
    ShadeInputs in;
    uint32_t i0, i1, i2;
    decode_triangle(sv, tri, i0, i1, i2, in.material);
    const uint32_t last = vertex_count - 1u;
    i0 = i0 < last ? i0 : last; i1 = i1 < last ? i1 : last; i2 = i2 < last ? i2 : last;
    const f3 p0 = load3(sv.positions, i0), p1 = load3(sv.positions, i1), p2 = load3(sv.positions, i2);
    in.geometry_normal = normalize(cross(p1 - p0, p2 - p0));
    in.n0 = load3(sv.normals, i0); in.n1 = load3(sv.normals, i1); in.n2 = load3(sv.normals, i2);
    in.uv0.x = sv.uvs[i0 * 2u]; in.uv0.y = sv.uvs[i0 * 2u + 1u];
    in.uv1.x = sv.uvs[i1 * 2u]; in.uv1.y = sv.uvs[i1 * 2u + 1u];
    in.uv2.x = sv.uvs[i2 * 2u]; in.uv2.y = sv.uvs[i2 * 2u + 1u];
    store_shade_record(records + (size_t)tri * kShadeRecordVec4, in);
}

hipError_t launch_wf_build_shade_records(hipStream_t s, const SceneView& sv, uint32_t triangle_count, uint32_t vertex_count, float4* records) {
    if (triangle_count == 0 || vertex_count == 0) return hipSuccess;
    hipLaunchKernelGGL(ptr_wf_build_shade_records, dim3((triangle_count + 255u) / 256u), dim3(256), 0, s, sv, triangle_count, vertex_count, records);
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
