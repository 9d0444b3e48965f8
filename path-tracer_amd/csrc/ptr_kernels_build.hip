// GPU construction of the flattened 4-wide BVH (the layout of host/bvh.hpp, version 2) straight from the scene's
// device buffers: Morton order -> binary tree (locally-ordered clustering, or the radix tree of Karras 2012) -> bottom-up
// boxes -> top-down collapse to 4-wide nodes with quantised child boxes.  It trades tree quality for build time
// (milliseconds instead of a second per few million triangles on the host's binned-SAH builder); the traversal kernels
// cannot tell the two apart.
//
//   1. per triangle: box and centroid; scene box by block reduction + ordered-integer atomics
//   2. 63-bit Morton key of the centroid (21 bits per axis), radix sort of (key, triangle) pairs (hipCUB)
//   3. binary tree: clusters of neighbours in Morton order merged round by round (3b below; the triangles are then put in
//      the tree's depth-first order), or the radix tree over the sorted keys, ties broken by position; parents recorded
//   4. boxes bottom-up: the second thread to reach a node merges its children and goes on
//   5. leaves: a subtree of at most 4 triangles becomes one leaf when that is cheaper, by the surface-area heuristic,
//      than entering its two children (evaluated bottom-up together with the boxes; the triangles of a subtree are
//      neighbours in the triangle order, so the records are simply written in that order); wide nodes level by level from
//      the root: a node adopts its two children and keeps replacing the inner child of largest surface area by that
//      child's children until it has four
//   6. the wide nodes are put in the order of the binary nodes they grew from (deterministic: along the Morton curve for the
//      radix tree; the top of the tree first and each clustering round along the curve for the clustered one)
#pragma clang fp contract(off)

#include <hipcub/hipcub.hpp>

#include <cstring>

#include "ptr_kernels.h"

namespace ptr {

namespace {

constexpr int32_t kEmptyChildRef = INT32_MIN;
constexpr uint32_t kLeafMax = 4;
constexpr uint32_t kClusterRoundsMax = 2048;  // clustering rounds before the build gives up (a round merges at least one pair; real scenes need tens)

struct BuildNode {      // one internal node of the binary radix tree
    int32_t left, right;       // >= 0: internal node, < 0: ~sorted position of a single triangle
    uint32_t first, last;      // range of sorted positions covered
    float lo[3], hi[3];
    float cost;                // surface-area cost of the cheapest way to finish this subtree (a leaf, or its two children)
    uint32_t leaf;             // 1: the subtree becomes one leaf
};

struct Frontier { uint32_t node, wide, pending; };

// order-preserving float <-> uint mapping for atomicMin / atomicMax
PTR_DEV uint32_t ordered(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
PTR_DEV float unordered(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

PTR_DEV void triangle_corners(const SceneView& sv, uint32_t tri, uint32_t vertex_count, f3& p0, f3& p1, f3& p2) {
    uint32_t i0, i1, i2, mat;
    decode_triangle(sv, tri, i0, i1, i2, mat);
    const uint32_t last = vertex_count - 1u;
    i0 = i0 < last ? i0 : last; i1 = i1 < last ? i1 : last; i2 = i2 < last ? i2 : last;
    p0 = load3(sv.positions, i0); p1 = load3(sv.positions, i1); p2 = load3(sv.positions, i2);
}

// ---- 1. boxes + scene bounds ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void build_boxes(SceneView sv, uint32_t n, uint32_t vertex_count, float4* box_lo, float4* box_hi, uint32_t* scene) {
    __shared__ uint32_t red[6];
    if (threadIdx.x < 3) red[threadIdx.x] = 0xffffffffu; else if (threadIdx.x < 6) red[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) {
        f3 a, b, c;
        triangle_corners(sv, i, vertex_count, a, b, c);
        const f3 lo = mk3(min_num(min_num(a.x, b.x), c.x), min_num(min_num(a.y, b.y), c.y), min_num(min_num(a.z, b.z), c.z));
        const f3 hi = mk3(max_num(max_num(a.x, b.x), c.x), max_num(max_num(a.y, b.y), c.y), max_num(max_num(a.z, b.z), c.z));
        box_lo[i] = make_float4(lo.x, lo.y, lo.z, 0.f);
        box_hi[i] = make_float4(hi.x, hi.y, hi.z, 0.f);
        atomicMin(&red[0], ordered(lo.x)); atomicMin(&red[1], ordered(lo.y)); atomicMin(&red[2], ordered(lo.z));
        atomicMax(&red[3], ordered(hi.x)); atomicMax(&red[4], ordered(hi.y)); atomicMax(&red[5], ordered(hi.z));
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&scene[threadIdx.x], red[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(&scene[threadIdx.x], red[threadIdx.x]);
}

// ---- 2. Morton keys ----------------------------------------------------------------------------------------
// 63 key bits, handed out from the top: each bit halves the axis along which the cells are currently longest, so the
// cells stay as close to cubes as the scene box allows (plain x, y, z interleaving halves a flat scene's short axis as
// often as its long ones, and every such split leaves both children with nearly the parent's surface area).
// An axis contributes at most 24 bits, what an fp32 coordinate resolves.
__global__ __launch_bounds__(256) void build_keys(uint32_t n, const float4* box_lo, const float4* box_hi, const uint32_t* scene, uint64_t* keys, uint32_t* vals) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 lo = box_lo[i], hi = box_hi[i];
    const float smin[3] = { unordered(scene[0]), unordered(scene[1]), unordered(scene[2]) };
    const float smax[3] = { unordered(scene[3]), unordered(scene[4]), unordered(scene[5]) };
    const float c[3] = { 0.5f * lo.x + 0.5f * hi.x, 0.5f * lo.y + 0.5f * hi.y, 0.5f * lo.z + 0.5f * hi.z };
    uint32_t q[3];
    float cell[3];
    for (int a = 0; a < 3; a++) {
        const float extent = smax[a] - smin[a];
        float t = extent > 0.f ? (c[a] - smin[a]) / extent : 0.f;
        t = t >= 0.f ? t : 0.f;                       // also catches NaN
        const float s = t * 16777216.0f;
        q[a] = s >= 16777215.0f ? 16777215u : (uint32_t)s;
        cell[a] = extent > 0.f ? extent : 0.f;
    }
    uint32_t used[3] = { 0u, 0u, 0u };
    uint64_t key = 0;
    for (int k = 0; k < 63; k++) {
        int a = -1;
        for (int j = 0; j < 3; j++)
            if (used[j] < 24u && (a < 0 || cell[j] > cell[a])) a = j;
        key = (key << 1) | (uint64_t)((q[a] >> (23u - used[a])) & 1u);
        used[a]++;
        cell[a] *= 0.5f;
    }
    keys[i] = key;
    vals[i] = i;
}

// ---- 3. radix tree -----------------------------------------------------------------------------------------
// common-prefix length of the (key, position) pairs at sorted positions i and j; -1 when j is out of range
PTR_DEV int prefix(const uint64_t* keys, uint32_t n, int64_t i, int64_t j) {
    if (j < 0 || j >= (int64_t)n) return -1;
    const uint64_t a = keys[i], b = keys[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz((int)((uint32_t)i ^ (uint32_t)j));
}
__global__ __launch_bounds__(256) void build_tree(uint32_t n, const uint64_t* keys, BuildNode* nodes, int32_t* node_parent, int32_t* leaf_parent) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)n - 1) return;
    const int d = prefix(keys, n, i, i + 1) - prefix(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = prefix(keys, n, i, i - d);
    int64_t lmax = 2;
    while (prefix(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int64_t l = 0;
    for (int64_t t = lmax / 2; t >= 1; t /= 2)
        if (prefix(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int64_t j = i + l * d;
    const int dnode = prefix(keys, n, i, j);
    int64_t s = 0;
    for (int64_t t = l;;) {
        t = (t + 1) / 2;
        if (prefix(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
    const int64_t gamma = i + s * d + (d < 0 ? -1 : 0);
    const int64_t first = i < j ? i : j, last = i < j ? j : i;
    BuildNode& nd = nodes[i];
    nd.first = (uint32_t)first; nd.last = (uint32_t)last;
    if (first == gamma) { nd.left = ~(int32_t)gamma; leaf_parent[gamma] = (int32_t)i; }
    else { nd.left = (int32_t)gamma; node_parent[gamma] = (int32_t)i; }
    if (last == gamma + 1) { nd.right = ~(int32_t)(gamma + 1); leaf_parent[gamma + 1] = (int32_t)i; }
    else { nd.right = (int32_t)(gamma + 1); node_parent[gamma + 1] = (int32_t)i; }
    if (i == 0) node_parent[0] = -1;
}

// ---- 3b. the binary tree by clustering instead of by key prefixes ---------------------------------------------
// Parallel locally-ordered clustering (Meister & Bittner 2018): the triangles start as one cluster each, in Morton order.
// Every round each cluster looks `radius` places to either side for the neighbour whose union with it has the smallest
// surface area; two clusters that choose each other merge into a node, which takes the place of the lower one; the array
// is compacted and the next round starts.  The tree follows the geometry instead of the bits of the keys: fewer node
// visits per ray than the radix tree for a few more milliseconds of build.
// The choice is made under a strict order of the candidate pairs that both ends agree on -- (area, distance in the array,
// a parity bit that differs between a cluster's two neighbours at one distance) -- so the pair that is smallest in the whole
// array always chooses each other: every round merges at least one pair, and equal boxes pair up instead of all choosing
// the same neighbour.
PTR_DEV float union_area(const float4 alo, const float4 ahi, const float4 blo, const float4 bhi) {
    const float dx = max_num(ahi.x, bhi.x) - min_num(alo.x, blo.x);
    const float dy = max_num(ahi.y, bhi.y) - min_num(alo.y, blo.y);
    const float dz = max_num(ahi.z, bhi.z) - min_num(alo.z, blo.z);
    const float a = dx * dy + dy * dz + dz * dx;
    return a == a ? a : __builtin_inff();
}
__global__ __launch_bounds__(256) void ploc_init(uint32_t n, const uint32_t* vals, const float4* box_lo, const float4* box_hi, int32_t* cluster, float4* lo, float4* hi) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n) return;
    const uint32_t t = vals[p];
    cluster[p] = ~(int32_t)p;
    lo[p] = box_lo[t];
    hi[p] = box_hi[t];
}
__global__ __launch_bounds__(256) void ploc_nearest(uint32_t m, uint32_t radius, const float4* lo, const float4* hi, uint32_t* nearest) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= m) return;
    const float4 alo = lo[i], ahi = hi[i];
    float best_area = __builtin_inff();
    uint32_t best_delta = 0xffffffffu, best_parity = 2u, best = i;
    for (uint32_t delta = 1; delta <= radius; delta++) {
        for (uint32_t side = 0; side < 2u; side++) {
            if (side == 0u ? i < delta : i + delta >= m) continue;
            const uint32_t j = side == 0u ? i - delta : i + delta;
            const float area = union_area(alo, ahi, lo[j], hi[j]);
            const uint32_t parity = ((side == 0u ? j : i) / delta) & 1u;
            if (area < best_area || (area == best_area && (delta < best_delta || (delta == best_delta && parity < best_parity)))) {
                best_area = area; best_delta = delta; best_parity = parity; best = j;
            }
        }
    }
    nearest[i] = best;
}
// per cluster: (stays in the array) << 32 | (is the lower end of a merging pair); the exclusive sums of both halves give the
// cluster's next position and the running number of the node it creates
__global__ __launch_bounds__(256) void ploc_flags(uint32_t m, const uint32_t* nearest, uint64_t* flags) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= m) return;
    const uint32_t j = nearest[i];
    const bool mutual = j != i && nearest[j] == i;
    flags[i] = ((uint64_t)(mutual && i > j ? 0u : 1u) << 32) | (uint64_t)(mutual && i < j ? 1u : 0u);
}
// Nodes are numbered downwards from n - 2 in the order they are created, so the last one, the root, is node 0.
// A node keeps the number of triangles below it in `last` until the triangle order is known.
__global__ __launch_bounds__(256) void ploc_merge(uint32_t m, uint32_t n, uint32_t created, const uint32_t* nearest, const uint64_t* flags, const uint64_t* sums,
                                                   const int32_t* cluster_in, const float4* lo_in, const float4* hi_in, int32_t* cluster_out, float4* lo_out,
                                                   float4* hi_out, BuildNode* nodes, int32_t* node_parent, int32_t* leaf_parent, uint64_t* totals) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= m) return;
    const uint64_t f = flags[i], before = sums[i];
    if (i == m - 1u) *totals = before + f;
    if ((f >> 32) == 0u) return;
    const uint32_t pos = (uint32_t)(before >> 32);
    if ((f & 1u) == 0u) {
        cluster_out[pos] = cluster_in[i]; lo_out[pos] = lo_in[i]; hi_out[pos] = hi_in[i];
        return;
    }
    const uint32_t j = nearest[i];
    const int32_t id = (int32_t)((n - 2u) - (created + (uint32_t)before));
    const int32_t a = cluster_in[i], b = cluster_in[j];
    const float4 alo = lo_in[i], ahi = hi_in[i], blo = lo_in[j], bhi = hi_in[j];
    const float4 ulo = make_float4(min_num(alo.x, blo.x), min_num(alo.y, blo.y), min_num(alo.z, blo.z), 0.f);
    const float4 uhi = make_float4(max_num(ahi.x, bhi.x), max_num(ahi.y, bhi.y), max_num(ahi.z, bhi.z), 0.f);
    BuildNode& nd = nodes[id];
    nd.left = a; nd.right = b;
    nd.first = 0u;
    nd.last = (a < 0 ? 1u : nodes[a].last) + (b < 0 ? 1u : nodes[b].last);
    nd.lo[0] = ulo.x; nd.lo[1] = ulo.y; nd.lo[2] = ulo.z; nd.hi[0] = uhi.x; nd.hi[1] = uhi.y; nd.hi[2] = uhi.z;
    nd.cost = 0.f; nd.leaf = 0u;
    if (a < 0) leaf_parent[(uint32_t)~a] = id; else node_parent[a] = id;
    if (b < 0) leaf_parent[(uint32_t)~b] = id; else node_parent[b] = id;
    cluster_out[pos] = id; lo_out[pos] = ulo; hi_out[pos] = uhi;
}
// The triangles of a subtree are not neighbours in Morton order any more; they are in the depth-first order of the tree.
// Position of a leaf in that order: the triangles of every left sibling subtree on its way to the root.
__global__ __launch_bounds__(256) void ploc_order(uint32_t n, const BuildNode* nodes, const int32_t* node_parent, const int32_t* leaf_parent, uint32_t* position) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n) return;
    int32_t ref = ~(int32_t)p, parent = leaf_parent[p];
    uint32_t pos = 0u;
    while (parent >= 0) {
        const BuildNode& nd = nodes[parent];
        if (nd.right == ref) pos += nd.left < 0 ? 1u : nodes[nd.left].last;
        ref = parent;
        parent = node_parent[parent];
    }
    position[p] = pos;
}
// every node's range [first, last] in the new order: set by its leftmost leaf, the only one that reaches it through left links alone
__global__ __launch_bounds__(256) void ploc_ranges(uint32_t n, BuildNode* nodes, const int32_t* node_parent, const int32_t* leaf_parent, const uint32_t* position) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n) return;
    const uint32_t first = position[p];
    int32_t ref = ~(int32_t)p, parent = leaf_parent[p];
    while (parent >= 0 && nodes[parent].left == ref) {
        const uint32_t count = nodes[parent].last;
        nodes[parent].first = first;
        nodes[parent].last = first + count - 1u;
        ref = parent;
        parent = node_parent[parent];
    }
}
// leaf references, the triangle list and the leaves' parents go over to the new order
__global__ __launch_bounds__(256) void ploc_reorder(uint32_t n, BuildNode* nodes, const uint32_t* position, const uint32_t* vals, const int32_t* leaf_parent,
                                                     uint32_t* vals_out, int32_t* leaf_parent_out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) {
        vals_out[position[i]] = vals[i];
        leaf_parent_out[position[i]] = leaf_parent[i];
    }
    if (i + 1u < n) {
        BuildNode& nd = nodes[i];
        if (nd.left < 0) nd.left = ~(int32_t)position[(uint32_t)~nd.left];
        if (nd.right < 0) nd.right = ~(int32_t)position[(uint32_t)~nd.right];
    }
}

// ---- 4. boxes bottom-up + triangle records (written in sorted order) ---------------------------------------
__global__ __launch_bounds__(256) void build_fit(SceneView sv, uint32_t n, uint32_t vertex_count, const uint32_t* vals, const float4* box_lo,
                                                 const float4* box_hi, BuildNode* nodes, const int32_t* node_parent, const int32_t* leaf_parent,
                                                 uint32_t* visits, float4* out_tris, float traversal_cost) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n) return;
    const uint32_t tri = vals[p];
    {
        f3 a, b, c;
        triangle_corners(sv, tri, vertex_count, a, b, c);
        const f3 e1 = b - a, e2 = c - a;
        const f3 nrm = cross(e1, e2);  // = (e1.y*e2.z - e2.y*e1.z, ...), the shader's operation order (ray.h:142-146)
        float4* rec = out_tris + 4 * (size_t)p;
        rec[0] = make_float4(a.x, a.y, a.z, __int_as_float((int32_t)tri));   // rank = primitive id: no reference order on the GPU
        rec[1] = make_float4(e1.x, e1.y, e1.z, __int_as_float((int32_t)tri));
        rec[2] = make_float4(e2.x, e2.y, e2.z, 0.f);
        rec[3] = make_float4(nrm.x, nrm.y, nrm.z, 0.f);
    }
    if (n == 1) return;
    int32_t node = leaf_parent[p];
    while (node >= 0) {
        __threadfence();
        if (atomicAdd(&visits[node], 1u) == 0u) return;  // the sibling subtree is not finished: its thread will continue
        __threadfence();
        BuildNode& nd = nodes[node];
        float lo[3], hi[3];
        float children_cost = 0.f;   // sum over the two children of area * cost
        for (int side = 0; side < 2; side++) {
            const int32_t c = side ? nd.right : nd.left;
            float clo[3], chi[3];
            float ccost = 1.f;       // one triangle test
            if (c < 0) {
                const uint32_t t = vals[(uint32_t)~c];
                const float4 l4 = box_lo[t], h4 = box_hi[t];
                clo[0] = l4.x; clo[1] = l4.y; clo[2] = l4.z; chi[0] = h4.x; chi[1] = h4.y; chi[2] = h4.z;
            } else {
                const volatile BuildNode& cn = nodes[c];
                for (int a = 0; a < 3; a++) { clo[a] = cn.lo[a]; chi[a] = cn.hi[a]; }
                ccost = cn.cost;
            }
            const float dx = chi[0] - clo[0], dy = chi[1] - clo[1], dz = chi[2] - clo[2];
            children_cost += (dx * dy + dy * dz + dz * dx) * ccost;
            for (int a = 0; a < 3; a++) {
                lo[a] = side ? min_num(lo[a], clo[a]) : clo[a];
                hi[a] = side ? max_num(hi[a], chi[a]) : chi[a];
            }
        }
        for (int a = 0; a < 3; a++) { nd.lo[a] = lo[a]; nd.hi[a] = hi[a]; }
        {
            const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
            const float area = dx * dy + dy * dz + dz * dx;
            const uint32_t count = nd.last - nd.first + 1u;
            const float split_cost = area > 0.f ? traversal_cost + children_cost / area : __builtin_inff();
            const bool leaf = count <= kLeafMax && !(split_cost < (float)count);
            nd.cost = leaf ? (float)count : split_cost;
            nd.leaf = leaf ? 1u : 0u;
        }
        node = node_parent[node];
    }
}

// ---- 5. collapse + quantise --------------------------------------------------------------------------------
struct ChildBox { float lo[3], hi[3]; };

PTR_DEV float step_from_biased(int biased) { return __uint_as_float((uint32_t)biased << 23); }

// Origin, power-of-two step and quantised [lo, hi] of one axis; the same rule as the host flattener:
// decoded lo <= true lo - margin and decoded hi >= true hi + margin, margin = 2^-20 of the node extent.
PTR_DEV bool quantise_axis(float node_lo, float node_hi, const float* lo, const float* hi, uint32_t count, float& origin, float& step, uint32_t& qlo, uint32_t& qhi) {
    const float extent = node_hi - node_lo;
    const float margin = max_num(extent * 9.5367431640625e-07f, 1e-30f);
    origin = nextafterf(node_lo - margin, -__builtin_inff());
    const float span = (node_hi + margin) - origin;
    int e;
    {
        const double per = (double)span / 255.0;
        int ex = 0;
        (void)frexp(per, &ex);                      // per = m * 2^ex, m in [0.5, 1)
        e = ex < -126 ? -126 : (ex > 127 ? 127 : ex);
    }
    for (;; e++) {
        if (e > 127) return false;
        const float st = step_from_biased(e + 127);
        bool ok = true;
        uint32_t pl = 0, ph = 0;
        for (uint32_t k = 0; k < count && ok; k++) {
            const float want_lo = lo[k] - margin, want_hi = hi[k] + margin;
            double ql = floor(((double)want_lo - (double)origin) / (double)st);
            ql = ql < 0.0 ? 0.0 : (ql > 255.0 ? 255.0 : ql);
            uint32_t a = (uint32_t)ql;
            while (a > 0u && origin + (float)a * st > want_lo) a--;
            double qh = ceil(((double)want_hi - (double)origin) / (double)st);
            qh = qh < 0.0 ? 0.0 : (qh > 255.0 ? 255.0 : qh);
            uint32_t b = (uint32_t)qh;
            while (b < 255u && origin + (float)b * st < want_hi) b++;
            if (origin + (float)a * st > want_lo || origin + (float)b * st < want_hi) { ok = false; break; }
            pl |= a << (8u * k); ph |= b << (8u * k);
        }
        if (ok) { step = st; qlo = pl; qhi = ph; return true; }
    }
}

// One wide node per frontier entry.  counters: [0] wide nodes allocated, [1] next frontier length, [2] stack need, [3] failure flag.
__global__ __launch_bounds__(128) void build_collapse(uint32_t count, const Frontier* in, Frontier* out, const BuildNode* nodes, const uint32_t* vals,
                                                      const float4* box_lo, const float4* box_hi, float4* out_nodes, uint32_t* counters,
                                                      uint32_t node_capacity, uint32_t* wide_of) {
    const uint32_t f = blockIdx.x * 128u + threadIdx.x;
    if (f >= count) return;
    const Frontier item = in[f];
    auto is_inner = [&](int32_t c) { return c >= 0 && nodes[c].leaf == 0u; };
    auto box_of = [&](int32_t c, ChildBox& b) {
        if (c < 0) {
            const uint32_t t = vals[(uint32_t)~c];
            const float4 l4 = box_lo[t], h4 = box_hi[t];
            b.lo[0] = l4.x; b.lo[1] = l4.y; b.lo[2] = l4.z; b.hi[0] = h4.x; b.hi[1] = h4.y; b.hi[2] = h4.z;
        } else {
            for (int a = 0; a < 3; a++) { b.lo[a] = nodes[c].lo[a]; b.hi[a] = nodes[c].hi[a]; }
        }
    };
    const BuildNode& self = nodes[item.node];
    int32_t child[4] = { self.left, self.right, 0, 0 };
    uint32_t n_children = 2;
    while (n_children < 4u) {
        int best = -1;
        float best_area = -1.f;
        for (uint32_t k = 0; k < n_children; k++) {
            if (!is_inner(child[k])) continue;
            const BuildNode& c = nodes[child[k]];
            const float dx = c.hi[0] - c.lo[0], dy = c.hi[1] - c.lo[1], dz = c.hi[2] - c.lo[2];
            const float area = 2.f * (dx * dy + dy * dz + dz * dx);
            if (area > best_area) { best_area = area; best = (int)k; }
        }
        if (best < 0) break;
        const BuildNode& c = nodes[child[best]];
        child[best] = c.left;
        child[n_children++] = c.right;
    }
    const uint32_t pending = item.pending + (n_children - 1u);
    atomicMax(&counters[2], pending);

    ChildBox boxes[4];
    int32_t refs[4] = { kEmptyChildRef, kEmptyChildRef, kEmptyChildRef, kEmptyChildRef };
    for (uint32_t k = 0; k < n_children; k++) {
        box_of(child[k], boxes[k]);
        if (is_inner(child[k])) {
            const uint32_t w = atomicAdd(&counters[0], 1u);
            if (w >= node_capacity) { counters[3] = 1u; return; }
            const uint32_t slot = atomicAdd(&counters[1], 1u);
            Frontier next; next.node = (uint32_t)child[k]; next.wide = w; next.pending = pending;
            out[slot] = next;
            refs[k] = (int32_t)w;
        } else if (child[k] < 0) {
            refs[k] = ~(int32_t)((((uint32_t)~child[k]) << 2) | 0u);
        } else {
            const BuildNode& c = nodes[child[k]];
            refs[k] = ~(int32_t)((c.first << 2) | (c.last - c.first));
        }
    }
    float origin[3], step[3];
    uint32_t qlo[3], qhi[3];
    for (int a = 0; a < 3; a++) {
        float lo[4], hi[4];
        for (uint32_t k = 0; k < n_children; k++) { lo[k] = boxes[k].lo[a]; hi[k] = boxes[k].hi[a]; }
        if (!quantise_axis(self.lo[a], self.hi[a], lo, hi, n_children, origin[a], step[a], qlo[a], qhi[a])) { counters[3] = 2u; return; }
    }
    wide_of[item.wide] = item.node;
    float4* nd = out_nodes + 4 * (size_t)item.wide;
    nd[0] = make_float4(origin[0], origin[1], origin[2], step[0]);
    nd[1] = make_float4(__int_as_float(refs[0]), __int_as_float(refs[1]), __int_as_float(refs[2]), __int_as_float(refs[3]));
    nd[2] = make_float4(__uint_as_float(qlo[0]), __uint_as_float(qlo[1]), __uint_as_float(qlo[2]), __uint_as_float(qhi[0]));
    nd[3] = make_float4(__uint_as_float(qhi[1]), __uint_as_float(qhi[2]), step[1], step[2]);
}

// ---- 6. memory order of the wide nodes ---------------------------------------------------------------------
// The level loop numbers nodes through an atomic counter: breadth first, in no particular order within a level, and
// differently from run to run.  The final order is that of the radix-tree nodes they grew from, which follows the
// Morton curve: neighbours in space are neighbours in memory, and the same scene always gives the same blob.
__global__ __launch_bounds__(256) void build_rank(uint32_t count, const uint32_t* old_of_new, uint32_t* new_of_old) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < count) new_of_old[old_of_new[i]] = i;
}
__global__ __launch_bounds__(256) void build_relayout(uint32_t count, const uint32_t* old_of_new, const uint32_t* new_of_old, const float4* in, float4* out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const float4* src = in + 4 * (size_t)old_of_new[i];
    float4* dst = out + 4 * (size_t)i;
    const float4 refs = src[1];
    int32_t r[4] = { __float_as_int(refs.x), __float_as_int(refs.y), __float_as_int(refs.z), __float_as_int(refs.w) };
    for (int k = 0; k < 4; k++) if (r[k] >= 0) r[k] = (int32_t)new_of_old[r[k]];
    dst[0] = src[0];
    dst[1] = make_float4(__int_as_float(r[0]), __int_as_float(r[1]), __int_as_float(r[2]), __int_as_float(r[3]));
    dst[2] = src[2];
    dst[3] = src[3];
}

// ---- ranks from the reference's threaded tree ---------------------------------------------------------------
// The reference keeps the later-visited triangle on equal t (ray.h:153); its visiting order is the order of the leaves in
// the threaded node array.  rank[primitive] = number of leaves before the primitive's own.
__global__ __launch_bounds__(256) void rank_flags(const ThreadedNode* nodes, uint32_t node_count, uint32_t* flags) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < node_count) flags[i] = nodes[i].primitive_id >= 0 ? 1u : 0u;
}
__global__ __launch_bounds__(256) void rank_scatter(const ThreadedNode* nodes, uint32_t node_count, const uint32_t* leaves_before, int32_t* rank_of_prim,
                                                    uint32_t triangle_count) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= node_count) return;
    const int32_t prim = nodes[i].primitive_id;
    if (prim >= 0 && (uint32_t)prim < triangle_count) rank_of_prim[prim] = (int32_t)leaves_before[i];
}
__global__ __launch_bounds__(256) void rank_apply(float4* records, uint32_t record_count, const int32_t* rank_of_prim, uint32_t triangle_count) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= record_count) return;
    const int32_t prim = __float_as_int(records[4 * (size_t)t + 1].w);
    if (prim >= 0 && (uint32_t)prim < triangle_count && rank_of_prim[prim] >= 0) records[4 * (size_t)t].w = __int_as_float(rank_of_prim[prim]);
}

struct FlatHeader {
    uint32_t magic, version, node_count, triangle_count;
    uint64_t nodes_offset, triangles_offset;
    int32_t root; uint32_t max_depth;
    float scene_min[3], scene_max[3];
};
static_assert(sizeof(FlatHeader) == 64, "header is one 64-byte line");

#define BUILD_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return e_; } } while (0)

}  // namespace

hipError_t apply_reference_ranks(hipStream_t s, const ThreadedNode* nodes, uint32_t node_count, float4* records, uint32_t record_count, uint32_t triangle_count) {
    if (node_count == 0 || record_count == 0 || triangle_count == 0) return hipSuccess;
    uint32_t *flags = nullptr, *before = nullptr;
    int32_t* rank_of_prim = nullptr;
    void* temp = nullptr;
    auto cleanup = [&]() { (void)hipFree(flags); (void)hipFree(before); (void)hipFree(rank_of_prim); (void)hipFree(temp); };
    hipError_t e;
    if ((e = hipMalloc((void**)&flags, (size_t)node_count * 4)) != hipSuccess) { cleanup(); return e; }
    if ((e = hipMalloc((void**)&before, (size_t)node_count * 4)) != hipSuccess) { cleanup(); return e; }
    if ((e = hipMalloc((void**)&rank_of_prim, (size_t)triangle_count * 4)) != hipSuccess) { cleanup(); return e; }
    if ((e = hipMemsetAsync(rank_of_prim, 0xff, (size_t)triangle_count * 4, s)) != hipSuccess) { cleanup(); return e; }   // -1: not in the tree, rank untouched
    const uint32_t nb = (node_count + 255u) / 256u;
    hipLaunchKernelGGL(rank_flags, dim3(nb), dim3(256), 0, s, nodes, node_count, flags);
    size_t bytes = 0;
    if ((e = hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, flags, before, (int)node_count, s)) != hipSuccess) { cleanup(); return e; }
    if ((e = hipMalloc(&temp, bytes ? bytes : 16)) != hipSuccess) { cleanup(); return e; }
    if ((e = hipcub::DeviceScan::ExclusiveSum(temp, bytes, flags, before, (int)node_count, s)) != hipSuccess) { cleanup(); return e; }
    hipLaunchKernelGGL(rank_scatter, dim3(nb), dim3(256), 0, s, nodes, node_count, before, rank_of_prim, triangle_count);
    hipLaunchKernelGGL(rank_apply, dim3((record_count + 255u) / 256u), dim3(256), 0, s, records, record_count, rank_of_prim, triangle_count);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    cleanup();
    return e;
}

size_t flat_build_blob_bytes(uint32_t triangle_count) {
    // header + at most one wide node per two triangles' worth of leaves... a wide node has >= 2 children, so there are
    // fewer wide nodes than leaves, and no more leaves than triangles
    return 64 + (size_t)triangle_count * 64 + (size_t)triangle_count * 64;
}

hipError_t build_flat_bvh(hipStream_t s, const SceneView& sv, uint32_t n, uint32_t vertex_count, void* blob, size_t blob_bytes, float traversal_cost, int cluster_radius, int* status) {
    *status = 0;
    if (n == 0 || vertex_count == 0 || blob_bytes < flat_build_blob_bytes(n)) { *status = -1; return hipSuccess; }
    const uint32_t node_capacity = n;
    void* pool[32] = {};
    int pooled = 0;
    auto cleanup = [&]() { for (int i = 0; i < pooled; i++) (void)hipFree(pool[i]); };
    auto take = [&](size_t bytes, void** out) -> hipError_t {
        if (pooled >= (int)(sizeof pool / sizeof pool[0])) return hipErrorOutOfMemory;
        hipError_t e = hipMalloc(out, bytes ? bytes : 16);
        if (e == hipSuccess) pool[pooled++] = *out;
        return e;
    };
    float4 *box_lo = nullptr, *box_hi = nullptr;
    uint64_t *keys = nullptr, *keys_sorted = nullptr;
    uint32_t *vals = nullptr, *vals_sorted = nullptr, *scene = nullptr, *visits = nullptr, *counters = nullptr;
    BuildNode* nodes = nullptr;
    int32_t *node_parent = nullptr, *leaf_parent = nullptr;
    Frontier* frontier[2] = { nullptr, nullptr };
    void* sort_temp = nullptr;
    BUILD_CHECK(take((size_t)n * 16, (void**)&box_lo));
    BUILD_CHECK(take((size_t)n * 16, (void**)&box_hi));
    BUILD_CHECK(take((size_t)n * 8, (void**)&keys));
    BUILD_CHECK(take((size_t)n * 8, (void**)&keys_sorted));
    BUILD_CHECK(take((size_t)n * 4, (void**)&vals));
    BUILD_CHECK(take((size_t)n * 4, (void**)&vals_sorted));
    BUILD_CHECK(take(64, (void**)&scene));
    BUILD_CHECK(take((size_t)n * sizeof(BuildNode), (void**)&nodes));
    BUILD_CHECK(take((size_t)n * 4 * 3, (void**)&node_parent));   // node_parent | leaf_parent | visits
    leaf_parent = node_parent + n;
    visits = (uint32_t*)(node_parent + 2 * (size_t)n);
    counters = scene + 8;

    const uint32_t init[16] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u, /* counters */ 1u, 0u, 0u, 0u, 0u, 0u, 0u, 0u };
    BUILD_CHECK(hipMemcpyAsync(scene, init, sizeof init, hipMemcpyHostToDevice, s));
    BUILD_CHECK(hipMemsetAsync(visits, 0, (size_t)n * 4, s));
    const uint32_t blocks = (n + 255u) / 256u;
    hipLaunchKernelGGL(build_boxes, dim3(blocks), dim3(256), 0, s, sv, n, vertex_count, box_lo, box_hi, scene);
    hipLaunchKernelGGL(build_keys, dim3(blocks), dim3(256), 0, s, n, box_lo, box_hi, scene, keys, vals);
    size_t temp_bytes = 0;
    BUILD_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, keys, keys_sorted, vals, vals_sorted, (int)n, 0, 63, s));
    BUILD_CHECK(take(temp_bytes, &sort_temp));
    BUILD_CHECK(hipcub::DeviceRadixSort::SortPairs(sort_temp, temp_bytes, keys, keys_sorted, vals, vals_sorted, (int)n, 0, 63, s));

    uint8_t* out = (uint8_t*)blob;
    float4* out_nodes = (float4*)(out + 64);
    float4* out_tris = (float4*)(out + 64 + (size_t)node_capacity * 64);
    if (n > 1 && cluster_radius <= 0) hipLaunchKernelGGL(build_tree, dim3((n - 1 + 255u) / 256u), dim3(256), 0, s, n, keys_sorted, nodes, node_parent, leaf_parent);
    if (n > 1 && cluster_radius > 0) {
        // the key arrays are free once the triangles are sorted: they hold the flags and their sums
        uint64_t* flags = keys;
        uint64_t* sums = keys_sorted;
        int32_t* cluster[2] = { nullptr, nullptr };
        float4 *clo[2] = { nullptr, nullptr }, *chi[2] = { nullptr, nullptr };
        uint32_t *nearest = nullptr, *position = nullptr, *vals_ordered = nullptr;
        int32_t* leaf_parent_ordered = nullptr;
        uint64_t* totals = nullptr;
        void* scan_temp = nullptr;
        for (int k = 0; k < 2; k++) {
            BUILD_CHECK(take((size_t)n * 4, (void**)&cluster[k]));
            BUILD_CHECK(take((size_t)n * 16, (void**)&clo[k]));
            BUILD_CHECK(take((size_t)n * 16, (void**)&chi[k]));
        }
        BUILD_CHECK(take((size_t)n * 4, (void**)&nearest));
        BUILD_CHECK(take((size_t)n * 4, (void**)&position));
        BUILD_CHECK(take((size_t)n * 4, (void**)&vals_ordered));
        BUILD_CHECK(take((size_t)n * 4, (void**)&leaf_parent_ordered));
        BUILD_CHECK(take(16, (void**)&totals));
        size_t scan_bytes = 0;
        BUILD_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, flags, sums, (int)n, s));
        BUILD_CHECK(take(scan_bytes, &scan_temp));
        hipLaunchKernelGGL(ploc_init, dim3(blocks), dim3(256), 0, s, n, vals_sorted, box_lo, box_hi, cluster[0], clo[0], chi[0]);
        uint32_t m = n, created = 0;
        int cur = 0;
        for (uint32_t round = 0; m > 1u; round++) {
            if (round >= kClusterRoundsMax) { *status = -6; cleanup(); return hipSuccess; }
            const uint32_t mb = (m + 255u) / 256u;
            hipLaunchKernelGGL(ploc_nearest, dim3(mb), dim3(256), 0, s, m, (uint32_t)cluster_radius, clo[cur], chi[cur], nearest);
            hipLaunchKernelGGL(ploc_flags, dim3(mb), dim3(256), 0, s, m, nearest, flags);
            size_t bytes = scan_bytes;
            BUILD_CHECK(hipcub::DeviceScan::ExclusiveSum(scan_temp, bytes, flags, sums, (int)m, s));
            hipLaunchKernelGGL(ploc_merge, dim3(mb), dim3(256), 0, s, m, n, created, nearest, flags, sums, cluster[cur], clo[cur], chi[cur], cluster[cur ^ 1],
                               clo[cur ^ 1], chi[cur ^ 1], nodes, node_parent, leaf_parent, totals);
            BUILD_CHECK(hipGetLastError());
            uint64_t host_totals = 0;
            BUILD_CHECK(hipMemcpyAsync(&host_totals, totals, sizeof host_totals, hipMemcpyDeviceToHost, s));
            BUILD_CHECK(hipStreamSynchronize(s));
            const uint32_t kept = (uint32_t)(host_totals >> 32), merged = (uint32_t)host_totals;
            if (merged == 0u || kept + merged != m) { *status = -6; cleanup(); return hipSuccess; }
            created += merged;
            m = kept;
            cur ^= 1;
        }
        if (created != n - 1u) { *status = -6; cleanup(); return hipSuccess; }
        const int32_t no_parent = -1;
        BUILD_CHECK(hipMemcpyAsync(node_parent, &no_parent, sizeof no_parent, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(ploc_order, dim3(blocks), dim3(256), 0, s, n, nodes, node_parent, leaf_parent, position);
        hipLaunchKernelGGL(ploc_ranges, dim3(blocks), dim3(256), 0, s, n, nodes, node_parent, leaf_parent, position);
        hipLaunchKernelGGL(ploc_reorder, dim3(blocks), dim3(256), 0, s, n, nodes, position, vals_sorted, leaf_parent, vals_ordered, leaf_parent_ordered);
        BUILD_CHECK(hipGetLastError());
        vals_sorted = vals_ordered;
        leaf_parent = leaf_parent_ordered;
    }
    hipLaunchKernelGGL(build_fit, dim3(blocks), dim3(256), 0, s, sv, n, vertex_count, vals_sorted, box_lo, box_hi, nodes, node_parent, leaf_parent, visits, out_tris, traversal_cost);
    BUILD_CHECK(hipGetLastError());

    FlatHeader hd{};
    hd.magic = 0x46525450u; hd.version = 2u; hd.triangle_count = n;
    hd.nodes_offset = 64; hd.triangles_offset = 64 + (uint64_t)node_capacity * 64;
    uint32_t host_scene[16];
    if (n <= kLeafMax) {
        hd.root = ~(int32_t)((0u << 2) | (n - 1u));
        hd.node_count = 0; hd.max_depth = 0;
        BUILD_CHECK(hipMemcpyAsync(host_scene, scene, sizeof host_scene, hipMemcpyDeviceToHost, s));
        BUILD_CHECK(hipStreamSynchronize(s));
    } else {
        // frontier buffers: at most one entry per wide node of a level; reuse the key arrays' storage is not possible
        // (keys_sorted is still read), so two fresh ones sized by the worst level (every leaf group's parent)
        BUILD_CHECK(take((size_t)(n / 2 + 2) * sizeof(Frontier), (void**)&frontier[0]));
        BUILD_CHECK(take((size_t)(n / 2 + 2) * sizeof(Frontier), (void**)&frontier[1]));
        const Frontier root{ 0u, 0u, 0u };
        BUILD_CHECK(hipMemcpyAsync(frontier[0], &root, sizeof root, hipMemcpyHostToDevice, s));
        uint32_t level_count = 1;
        int cur = 0;
        for (int level = 0; level_count > 0; level++) {
            if (level > 4096) { *status = -2; cleanup(); return hipSuccess; }
            BUILD_CHECK(hipMemsetAsync(counters + 1, 0, 4, s));
            hipLaunchKernelGGL(build_collapse, dim3((level_count + 127u) / 128u), dim3(128), 0, s, level_count, frontier[cur], frontier[cur ^ 1], nodes,
                               vals_sorted, box_lo, box_hi, out_nodes, counters, node_capacity, visits);
            BUILD_CHECK(hipGetLastError());
            BUILD_CHECK(hipMemcpyAsync(host_scene, scene, sizeof host_scene, hipMemcpyDeviceToHost, s));
            BUILD_CHECK(hipStreamSynchronize(s));
            if (host_scene[8 + 3] != 0u) { *status = -3; cleanup(); return hipSuccess; }
            level_count = host_scene[8 + 1];
            if (level_count > n / 2 + 2) { *status = -4; cleanup(); return hipSuccess; }
            cur ^= 1;
        }
        hd.root = 0;
        hd.node_count = host_scene[8 + 0];
        hd.max_depth = host_scene[8 + 2];
        if (hd.max_depth > kStackMax) { *status = -5; cleanup(); return hipSuccess; }
        {
            // order the wide nodes by the radix-tree node each grew from (visits[] holds that id since the level loop;
            // the arrays of the earlier stages are free again and serve as sort buffers).  The root grew from node 0 and stays first.
            const uint32_t wide = hd.node_count;
            uint32_t* wide_of = visits;
            uint32_t* sorted_of = (uint32_t*)node_parent;     // keys out
            uint32_t* old_of_new = (uint32_t*)leaf_parent;    // values out: the old index of every new position
            uint32_t* new_of_old = (uint32_t*)keys;
            size_t bytes = 0;
            BUILD_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, wide_of, sorted_of, vals, old_of_new, (int)wide, 0, 32, s));
            void* temp = nullptr;
            BUILD_CHECK(take(bytes, &temp));
            BUILD_CHECK(hipcub::DeviceRadixSort::SortPairs(temp, bytes, wide_of, sorted_of, vals, old_of_new, (int)wide, 0, 32, s));
            float4* moved = nullptr;
            BUILD_CHECK(take((size_t)wide * 64, (void**)&moved));
            const uint32_t wb = (wide + 255u) / 256u;
            hipLaunchKernelGGL(build_rank, dim3(wb), dim3(256), 0, s, wide, old_of_new, new_of_old);
            hipLaunchKernelGGL(build_relayout, dim3(wb), dim3(256), 0, s, wide, old_of_new, new_of_old, out_nodes, moved);
            BUILD_CHECK(hipGetLastError());
            BUILD_CHECK(hipMemcpyAsync(out_nodes, moved, (size_t)wide * 64, hipMemcpyDeviceToDevice, s));
        }
    }
    // scene box: decode the ordered integers on the host
    for (int a = 0; a < 6; a++) {
        const uint32_t u = host_scene[a];
        const uint32_t bits = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
        float f;
        memcpy(&f, &bits, 4);
        (a < 3 ? hd.scene_min[a] : hd.scene_max[a - 3]) = f;
    }
    BUILD_CHECK(hipMemcpyAsync(out, &hd, sizeof hd, hipMemcpyHostToDevice, s));
    BUILD_CHECK(hipStreamSynchronize(s));
    cleanup();
    return hipSuccess;
}

}  // namespace ptr
