#include "bvh.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <thread>

namespace {

// ------------------------------------------------------------------------------------------------
// reference_order: the reference's bucketed top-down build (src/bvh.cpp:150-262), restated with an
// explicit work stack (the reference recurses once per node, as deep as the tree) and
// structure-of-arrays scratch.  What must be reproduced for bit-identical output:
//   * count 1 / 2 are leaves; count <= 4 sorts the range by centroid on the PARENT box's longest
//     axis and halves it; otherwise 12 centroid buckets on that axis, cost
//     0.125 + (nL*SA_L + nR*SA_R) / SA_parent in fp32, first strict minimum, and the split POSITION
//     is the number of primitives in buckets <= best -- applied to the range in its current order
//     (the range is not partitioned by bucket).
//   * child boxes are recomputed from the actual members of each half.
// ------------------------------------------------------------------------------------------------
struct ref_tree {
    struct node {
        aabb box;
        int32_t left = -1;  // right child = left + 1
        int32_t prim = -1;
    };
    std::vector<node> nodes;
};

constexpr uint32_t kRefBuckets = 12;

aabb range_bounds(const std::vector<aabb>& prim_box, const std::vector<int32_t>& order, uint32_t begin, uint32_t end) {
    aabb b;
    for (uint32_t i = begin; i < end; i++) b.union_with(prim_box[order[i]]);
    return b;
}

// Number of primitives that go left for a range of more than four primitives.
uint32_t ref_bucket_split(const std::vector<triangle>& tris, const std::vector<aabb>& prim_box,
                          const std::vector<int32_t>& order, uint32_t begin, uint32_t end, const aabb& parent_box,
                          int32_t axis) {
    uint32_t counts[kRefBuckets] = {};
    aabb boxes[kRefBuckets];

    aabb centroid_bounds;
    for (uint32_t i = begin; i < end; i++) centroid_bounds.union_with(tris[order[i]].center);

    const float extent = centroid_bounds.maximum[axis] - centroid_bounds.minimum[axis];
    if (!(extent > 0.f)) {
        throw bvh_build_error(
            "reference_order build does not terminate: every centroid of a >4-triangle range is equal on the split axis");
    }
    for (uint32_t i = begin; i < end; i++) {
        const float offset = (tris[order[i]].center[axis] - centroid_bounds.minimum[axis]) / extent;
        uint32_t b = (uint32_t)(offset * (float)kRefBuckets);
        if (b == kRefBuckets) b -= 1;
        counts[b]++;
        boxes[b].union_with(prim_box[order[i]]);
    }

    float cost[kRefBuckets - 1];
    const float parent_area = parent_box.surface_area();
    for (uint32_t s = 0; s + 1 < kRefBuckets; s++) {
        aabb lbox, rbox;
        float ln = 0.f, rn = 0.f;
        for (uint32_t i = 0; i <= s; i++) { lbox.union_with(boxes[i]); ln += (float)counts[i]; }
        for (uint32_t i = s + 1; i < kRefBuckets; i++) { rbox.union_with(boxes[i]); rn += (float)counts[i]; }
        const float l = ln * lbox.surface_area();
        const float r = rn * rbox.surface_area();
        cost[s] = 0.125f + (l + r) / parent_area;
    }
    uint32_t best = 0;
    for (uint32_t s = 1; s + 1 < kRefBuckets; s++) {
        if (cost[s] < cost[best]) best = s;
    }
    uint32_t left_count = 0;
    for (uint32_t i = 0; i <= best; i++) left_count += counts[i];
    return left_count;
}

ref_tree build_reference_order(const std::vector<triangle>& tris) {
    const uint32_t n = (uint32_t)tris.size();
    ref_tree t;
    t.nodes.resize(2 * (size_t)n - 1);

    std::vector<aabb> prim_box(n);
    std::vector<int32_t> order(n);
    for (uint32_t i = 0; i < n; i++) { prim_box[i] = aabb(tris[i]); order[i] = (int32_t)i; }

    auto make_leaf = [&](uint32_t id, int32_t prim) {
        t.nodes[id].box = prim_box[prim];
        t.nodes[id].prim = prim;
        t.nodes[id].left = -1;
    };

    struct work { uint32_t id, begin, end; };
    std::vector<work> stack;
    uint32_t next_id = 0;
    t.nodes[0].box = range_bounds(prim_box, order, 0, n);
    stack.push_back({ 0, 0, n });

    while (!stack.empty()) {
        const work w = stack.back();
        stack.pop_back();
        const uint32_t count = w.end - w.begin;
        if (count == 1) { make_leaf(w.id, order[w.begin]); continue; }

        const uint32_t left = ++next_id;
        const uint32_t right = ++next_id;
        t.nodes[w.id].left = (int32_t)left;
        if (count == 2) {
            make_leaf(left, order[w.begin]);
            make_leaf(right, order[w.begin + 1]);
            continue;
        }

        const aabb parent_box = t.nodes[w.id].box;
        const int32_t axis = parent_box.maximum_axis();
        uint32_t left_count;
        if (count <= 4) {
            // libstdc++'s std::sort on <= 16 elements is a (stable) insertion sort.
            std::stable_sort(order.begin() + w.begin, order.begin() + w.end,
                             [&](int32_t a, int32_t b) { return tris[a].center[axis] < tris[b].center[axis]; });
            left_count = count / 2;
        } else {
            left_count = ref_bucket_split(tris, prim_box, order, w.begin, w.end, parent_box, axis);
            if (left_count == 0 || left_count >= count) {
                throw bvh_build_error("reference_order build does not terminate: empty split");
            }
        }
        const uint32_t mid = w.begin + left_count;
        t.nodes[left].box = range_bounds(prim_box, order, w.begin, mid);
        t.nodes[right].box = range_bounds(prim_box, order, mid, w.end);
        stack.push_back({ right, mid, w.end });  // left subtree is numbered first
        stack.push_back({ left, w.begin, mid });
    }
    return t;
}

// Depth-first renumbering + skip links (src/bvh.cpp:264-279), then the 32-byte records.
void thread_tree(const ref_tree& t, std::vector<packed_bvh_node>& out) {
    const size_t n = t.nodes.size();
    out.assign(n, packed_bvh_node{});
    std::vector<int32_t> df(n, -1), skip(n, -1);

    struct visit { int32_t id, next; };
    std::vector<visit> stack;
    stack.push_back({ 0, -1 });
    int32_t counter = 0;
    while (!stack.empty()) {
        const visit v = stack.back();
        stack.pop_back();
        df[v.id] = counter++;
        skip[v.id] = v.next;
        const int32_t left = t.nodes[v.id].left;
        if (left > 0) {
            stack.push_back({ left + 1, v.next });
            stack.push_back({ left, left + 1 });
        }
    }
    for (size_t id = 0; id < n; id++) {
        const auto& src = t.nodes[id];
        packed_bvh_node& dst = out[df[id]];
        dst.primitive_id = src.prim;
        for (int a = 0; a < 3; a++) { dst.min[a] = src.box.minimum.v[a]; dst.max[a] = src.box.maximum.v[a]; }
        dst.next_id = skip[id] == -1 ? -1 : df[skip[id]];
    }
}

// ------------------------------------------------------------------------------------------------
// sah_partitioned: binned SAH over all three axes, primitives partitioned in place.
// ------------------------------------------------------------------------------------------------
struct sah_node {
    aabb box;
    int32_t left = -1, right = -1;  // indices into the same vector; -1/-1 = leaf
    uint32_t first = 0, count = 0;  // leaf range in `order`
};

// Bins per axis of a split: as many as the range has triangles, between 16 and 256.  (Node visits per ray on the benchmark
// scene: 19.98 with 16 bins everywhere, 19.54 with up to 64, 19.28 with up to 256, 19.32 with up to 1024 -- saturated; build
// time +4 % from 64 to 256.  Small ranges gain nothing from more bins than triangles.)
constexpr int kSahBinsMin = 16, kSahBinsMax = 256;
inline int sah_bins(uint32_t count) { return count < (uint32_t)kSahBinsMin ? kSahBinsMin : (count > (uint32_t)kSahBinsMax ? kSahBinsMax : (int)count); }
constexpr uint32_t kForceMedianDepth = 44;  // past this depth split at the object median: bounds the depth

struct sah_builder {
    const std::vector<triangle>& tris;
    const std::vector<aabb>& prim_box;
    std::vector<int32_t>& order;
    uint32_t leaf_max;
    float traversal_cost, triangle_cost;

    // Splits [begin, end) in place; returns the split point, or begin when the range should be a leaf.
    uint32_t split(uint32_t begin, uint32_t end, const aabb& box, uint32_t depth) const {
        const uint32_t count = end - begin;
        if (count <= 1) return begin;

        aabb cb;
        for (uint32_t i = begin; i < end; i++) cb.union_with(tris[order[i]].center);

        int best_axis = -1, best_bin = -1;
        float best_cost = std::numeric_limits<float>::infinity();
        const int bins = sah_bins(count);
        if (depth < kForceMedianDepth) {
            for (int axis = 0; axis < 3; axis++) {
                const float lo = cb.minimum.v[axis];
                const float extent = cb.maximum.v[axis] - lo;
                if (!(extent > 0.f)) continue;
                const float scale = (float)bins / extent;
                uint32_t bin_count[kSahBinsMax] = {};
                aabb bin_box[kSahBinsMax];
                for (uint32_t i = begin; i < end; i++) {
                    int b = (int)((tris[order[i]].center.v[axis] - lo) * scale);
                    b = b < 0 ? 0 : (b >= bins ? bins - 1 : b);
                    bin_count[b]++;
                    bin_box[b].union_with(prim_box[order[i]]);
                }
                float right_area[kSahBinsMax];
                uint32_t right_count[kSahBinsMax];
                aabb acc;
                uint32_t n = 0;
                for (int b = bins - 1; b > 0; b--) {
                    acc.union_with(bin_box[b]);
                    n += bin_count[b];
                    right_area[b] = n ? acc.surface_area() : 0.f;
                    right_count[b] = n;
                }
                acc = aabb();
                n = 0;
                for (int b = 0; b + 1 < bins; b++) {
                    acc.union_with(bin_box[b]);
                    n += bin_count[b];
                    if (n == 0 || right_count[b + 1] == 0) continue;
                    const float c = (float)n * acc.surface_area() + (float)right_count[b + 1] * right_area[b + 1];
                    if (c < best_cost) { best_cost = c; best_axis = axis; best_bin = b; }
                }
            }
        }

        const float area = box.surface_area();
        if (best_axis >= 0 && count <= leaf_max) {
            const float split_cost = traversal_cost + triangle_cost * best_cost / (area > 0.f ? area : 1.f);
            if (split_cost >= triangle_cost * (float)count) return begin;  // leaf is cheaper
        }

        if (best_axis >= 0) {
            const float lo = cb.minimum.v[best_axis];
            const float scale = (float)bins / (cb.maximum.v[best_axis] - lo);
            auto mid = std::partition(order.begin() + begin, order.begin() + end, [&](int32_t p) {
                int b = (int)((tris[p].center.v[best_axis] - lo) * scale);
                b = b < 0 ? 0 : (b >= bins ? bins - 1 : b);
                return b <= best_bin;
            });
            const uint32_t m = (uint32_t)(mid - order.begin());
            if (m > begin && m < end) return m;
        }
        // No usable SAH split (coincident centroids, or depth limit): object median on the longest axis.
        if (count <= leaf_max) return begin;
        const int axis = box.maximum_axis();
        const uint32_t m = begin + count / 2;
        std::nth_element(order.begin() + begin, order.begin() + m, order.begin() + end, [&](int32_t a, int32_t b) {
            const float ca = tris[a].center.v[axis], cb2 = tris[b].center.v[axis];
            return ca < cb2 || (ca == cb2 && a < b);
        });
        return m;
    }

    // Depth-first build of one subtree into `nodes`; nodes[root] must already hold its box.
    void build(std::vector<sah_node>& nodes, uint32_t root, uint32_t begin, uint32_t end, uint32_t depth) const {
        struct work { uint32_t id, begin, end, depth; };
        std::vector<work> stack{ { root, begin, end, depth } };
        while (!stack.empty()) {
            const work w = stack.back();
            stack.pop_back();
            const aabb box = nodes[w.id].box;
            const uint32_t m = split(w.begin, w.end, box, w.depth);
            if (m == w.begin) {
                nodes[w.id].first = w.begin;
                nodes[w.id].count = w.end - w.begin;
                continue;
            }
            const uint32_t l = (uint32_t)nodes.size();
            nodes.emplace_back();
            nodes.emplace_back();
            nodes[l].box = range_bounds(prim_box, order, w.begin, m);
            nodes[l + 1].box = range_bounds(prim_box, order, m, w.end);
            nodes[w.id].left = (int32_t)l;
            nodes[w.id].right = (int32_t)l + 1;
            stack.push_back({ l + 1, m, w.end, w.depth + 1 });
            stack.push_back({ l, w.begin, m, w.depth + 1 });
        }
    }
};

// Builds the SAH tree with `threads` workers: the top of the tree is split breadth-first on the
// calling thread until there are enough independent ranges, which are then built concurrently
// (each into its own node vector) and stitched back.
std::vector<sah_node> build_sah(const std::vector<triangle>& tris, std::vector<int32_t>& order, uint32_t leaf_max,
                                unsigned threads) {
    const uint32_t n = (uint32_t)tris.size();
    std::vector<aabb> prim_box(n);
    order.resize(n);
    for (uint32_t i = 0; i < n; i++) { prim_box[i] = aabb(tris[i]); order[i] = (int32_t)i; }

    // traversal and triangle costs: frame time on MI355X is flat for traversal costs 0.3 .. 1.5 and worse beyond
    sah_builder b{ tris, prim_box, order, leaf_max, 1.0f, 1.0f };
    std::vector<sah_node> nodes(1);
    nodes[0].box = range_bounds(prim_box, order, 0, n);

    if (threads == 0) threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (threads <= 1 || n < 65536) {
        b.build(nodes, 0, 0, n, 0);
        return nodes;
    }

    struct pending { uint32_t id, begin, end, depth; };
    std::vector<pending> frontier{ { 0, 0, n, 0 } };
    const uint32_t grain = std::max<uint32_t>(4096, n / (threads * 8));
    for (;;) {
        // split the largest frontier range until all are below the grain (or cannot be split)
        size_t big = frontier.size();
        uint32_t big_count = grain;
        for (size_t i = 0; i < frontier.size(); i++) {
            const uint32_t c = frontier[i].end - frontier[i].begin;
            if (c > big_count) { big = i; big_count = c; }
        }
        if (big == frontier.size() || frontier.size() >= 4096) break;
        const pending p = frontier[big];
        const uint32_t m = b.split(p.begin, p.end, nodes[p.id].box, p.depth);
        if (m == p.begin) {  // became a leaf
            nodes[p.id].first = p.begin;
            nodes[p.id].count = p.end - p.begin;
            frontier.erase(frontier.begin() + big);
            continue;
        }
        const uint32_t l = (uint32_t)nodes.size();
        nodes.emplace_back();
        nodes.emplace_back();
        nodes[l].box = range_bounds(prim_box, order, p.begin, m);
        nodes[l + 1].box = range_bounds(prim_box, order, m, p.end);
        nodes[p.id].left = (int32_t)l;
        nodes[p.id].right = (int32_t)l + 1;
        frontier[big] = { l, p.begin, m, p.depth + 1 };
        frontier.push_back({ l + 1, m, p.end, p.depth + 1 });
    }

    std::vector<std::vector<sah_node>> sub(frontier.size());
    std::atomic<size_t> next{ 0 };
    auto worker = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= frontier.size()) return;
            const pending& p = frontier[i];
            sub[i].assign(1, sah_node{});
            sub[i][0].box = nodes[p.id].box;
            b.build(sub[i], 0, p.begin, p.end, p.depth);
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; t++) pool.emplace_back(worker);
    for (auto& t : pool) t.join();

    for (size_t i = 0; i < frontier.size(); i++) {
        const uint32_t base = (uint32_t)nodes.size();  // sub[i][k], k >= 1, lands at base + k - 1
        auto remap = [&](int32_t k) { return k < 0 ? k : (int32_t)(base + (uint32_t)k - 1); };
        sah_node root = sub[i][0];
        root.left = remap(root.left);
        root.right = remap(root.right);
        nodes[frontier[i].id] = root;
        for (size_t k = 1; k < sub[i].size(); k++) {
            sah_node nd = sub[i][k];
            nd.left = remap(nd.left);
            nd.right = remap(nd.right);
            nodes.push_back(nd);
        }
    }
    return nodes;
}

void fill_flat_triangle(flat_bvh_triangle& out, const triangle& t, int32_t prim, int32_t rank) {
    const float* a = t.p1.v;
    const float* b = t.p2.v;
    const float* c = t.p3.v;
    float e1[3], e2[3];
    for (int i = 0; i < 3; i++) { e1[i] = b[i] - a[i]; e2[i] = c[i] - a[i]; out.p0[i] = a[i]; out.e1[i] = e1[i]; out.e2[i] = e2[i]; }
    out.n[0] = e1[1] * e2[2] - e2[1] * e1[2];
    out.n[1] = e1[2] * e2[0] - e2[2] * e1[0];
    out.n[2] = e1[0] * e2[1] - e2[0] * e1[1];
    out.rank = rank;
    out.primitive_id = prim;
    out.kind = FLAT_BVH_KIND_TRIANGLE;
    out.pad1 = 0;
}

void fill_flat_sphere(flat_bvh_triangle& out, const flat_bvh_sphere& s, int32_t index, int32_t rank) {
    std::memset(&out, 0, sizeof out);
    for (int i = 0; i < 3; i++) out.p0[i] = s.center[i];
    out.e1[0] = s.radius;
    out.rank = rank;
    out.primitive_id = FLAT_BVH_SPHERE_TAG | index;
    out.kind = FLAT_BVH_KIND_SPHERE;
}

// A sphere as the builder sees it: a "triangle" whose three points span the sphere's box (rounded outwards) and whose
// centre is the sphere's.
triangle sphere_proxy(const flat_bvh_sphere& s) {
    const float inf = std::numeric_limits<float>::infinity();
    float lo[3], hi[3];
    for (int i = 0; i < 3; i++) {
        lo[i] = std::nextafter(s.center[i] - s.radius, -inf);
        hi[i] = std::nextafter(s.center[i] + s.radius, inf);
    }
    return triangle(vec3(lo[0], lo[1], lo[2]), vec3(hi[0], hi[1], hi[2]), vec3(s.center[0], s.center[1], s.center[2]));
}

}  // namespace

bvh::bvh(std::vector<triangle>& tris, std::vector<packed_bvh_node>& packed_nodes)
    : bvh(tris, packed_nodes, build_mode::reference_order) {}

bvh::bvh(std::vector<triangle>& tris, std::vector<packed_bvh_node>& packed_nodes, build_mode mode)
    : triangles(tris), packed(&packed_nodes) {
    if (tris.empty()) throw bvh_build_error("bvh: no triangles");
    if (mode == build_mode::reference_order) {
        thread_tree(build_reference_order(tris), packed_nodes);
        return;
    }
    // sah_partitioned in the threaded layout: one triangle per leaf, same 32-byte records.
    std::vector<int32_t> order;
    const std::vector<sah_node> nodes = build_sah(tris, order, 1, 0);
    ref_tree t;
    t.nodes.resize(nodes.size());
    // children must be adjacent (right = left + 1) for the threaded numbering: build_sah allocates them so.
    for (size_t i = 0; i < nodes.size(); i++) {
        t.nodes[i].box = nodes[i].box;
        t.nodes[i].left = nodes[i].left;
        t.nodes[i].prim = nodes[i].left < 0 ? order[nodes[i].first] : -1;
    }
    thread_tree(t, packed_nodes);
}

void bvh::exporter() {
    std::cout << "digraph bvh {\n";
    if (packed) {
        // parent -> child edges recovered from the threaded array: an inner node's children are
        // id + 1 and the skip target of id + 1.
        const auto& nodes = *packed;
        for (size_t id = 0; id < nodes.size(); id++) {
            if (nodes[id].primitive_id != -1) continue;
            std::cout << "\t" << id << " -> " << id + 1 << ";\n";
            std::cout << "\t" << id << " -> " << nodes[id + 1].next_id << ";\n";
        }
    }
    std::cout << "}\n";
}

std::vector<int32_t> bvh::visit_ranks(const std::vector<packed_bvh_node>& packed_nodes, size_t primitive_count) {
    std::vector<int32_t> ranks(primitive_count, -1);
    int32_t r = 0;
    for (const auto& nd : packed_nodes) {
        if (nd.primitive_id >= 0 && (size_t)nd.primitive_id < primitive_count) ranks[nd.primitive_id] = r++;
    }
    return ranks;
}

namespace {

// ---- 4-wide collapse + 8-bit quantisation of the SAH tree ----------------------------------------------
struct wide_node {
    int32_t child[4];   // indices into the sah_node vector
    uint32_t count = 0;
};

// 2^(biased - 127) as a float, the per-axis quantisation step of a node
float scale_from_biased(uint32_t biased) {
    const uint32_t bits = biased << 23;
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

// Decoded bound exactly as the kernels compute it: origin + q * step (the product is exact: q <= 255 and step
// is a power of two, so a fused multiply-add and a multiply followed by an add give the same bits).
float dequantise(float origin, uint32_t q, float step) { return origin + (float)q * step; }

// Chooses the origin and step of one axis and quantises the children's [lo, hi] with a safety margin:
//     decoded lo <= true lo - margin   and   decoded hi >= true hi + margin,   margin = 2^-20 * node extent.
// The margin is what makes the kernels' folded slab arithmetic (t = fma(q, step/d, (origin - o)/d)) conservative:
// its absolute rounding error is a few 2^-24 of the node extent when the ray starts near the node, and relative
// to t (covered by the pad on the exit distance) when it starts far away.
void quantise_axis(float node_lo, float node_hi, const float* lo, const float* hi, uint32_t count, float& origin, uint32_t& biased,
                   uint8_t* qlo, uint8_t* qhi) {
    const float extent = node_hi - node_lo;
    const float margin = std::max(extent * 9.5367431640625e-07f, 1e-30f);  // 2^-20 * extent
    origin = std::nextafter(node_lo - margin, -std::numeric_limits<float>::infinity());
    const float span = (node_hi + margin) - origin;
    int e = -126;
    {
        int ex;
        std::frexp((double)span / 255.0, &ex);  // span/255 = m * 2^ex, m in [0.5, 1)  ->  2^ex >= span/255
        e = std::max(-126, std::min(127, ex));
    }
    for (;; e++) {
        if (e > 127) throw bvh_build_error("flatten: node extent cannot be quantised");
        const float step = scale_from_biased((uint32_t)(e + 127));
        bool ok = true;
        for (uint32_t k = 0; k < count && ok; k++) {
            const float want_lo = lo[k] - margin, want_hi = hi[k] + margin;
            double ql = std::floor(((double)want_lo - (double)origin) / (double)step);
            ql = std::max(0.0, std::min(255.0, ql));
            uint32_t a = (uint32_t)ql;
            while (a > 0 && dequantise(origin, a, step) > want_lo) a--;
            double qh = std::ceil(((double)want_hi - (double)origin) / (double)step);
            qh = std::max(0.0, std::min(255.0, qh));
            uint32_t b = (uint32_t)qh;
            while (b < 255 && dequantise(origin, b, step) < want_hi) b++;
            if (dequantise(origin, a, step) > want_lo || dequantise(origin, b, step) < want_hi) { ok = false; break; }
            qlo[k] = (uint8_t)a;
            qhi[k] = (uint8_t)b;
        }
        if (ok) { biased = (uint32_t)(e + 127); return; }
    }
}

}  // namespace

namespace {

// The SAH tree collapsed to 4 children per node and numbered in depth-first pre-order, with its leaves' items laid
// out in leaf order and the traversal stack it needs.
struct wide_layout {
    std::vector<wide_node> wide;
    std::vector<uint32_t> wide_of;      // sah id of each wide node
    std::vector<int32_t> wide_index;    // wide index of each sah inner node that survived the collapse
    std::vector<uint32_t> leaf_first;   // first item of each sah leaf, in leaf order
    uint32_t item_count = 0, stack_need = 0;
};

wide_layout collapse_tree(const std::vector<sah_node>& nodes) {
    // Collapse to 4 children per node: starting from a node's two children, the inner child with the largest
    // surface area is replaced by its own two children until there are four (or only leaves are left).
    auto collapse = [&](uint32_t id) {
        wide_node w;
        w.child[0] = nodes[id].left; w.child[1] = nodes[id].right; w.count = 2;
        while (w.count < 4) {
            int best = -1;
            float best_area = -1.f;
            for (uint32_t k = 0; k < w.count; k++) {
                const sah_node& c = nodes[w.child[k]];
                if (c.left < 0) continue;
                const float a = c.box.surface_area();
                if (a > best_area) { best_area = a; best = (int)k; }
            }
            if (best < 0) break;
            const sah_node& c = nodes[w.child[best]];
            w.child[best] = c.left;           // keep the near-origin order stable: left in place, right appended
            w.child[w.count++] = c.right;
        }
        return w;
    };

    // Depth-first pre-order: number the wide nodes, place the items in leaf order, bound the stack.
    wide_layout L;
    L.wide_index.assign(nodes.size(), -1);
    L.leaf_first.assign(nodes.size(), 0);
    if (nodes[0].left < 0) {
        L.leaf_first[0] = 0;
        L.item_count = nodes[0].count;
        return L;
    }
    struct visit { uint32_t id, pending; };  // pending = stack entries that can be live when this node is entered
    std::vector<visit> stack{ { 0, 0 } };
    while (!stack.empty()) {
        const visit v = stack.back();
        stack.pop_back();
        if (nodes[v.id].left < 0) {
            L.leaf_first[v.id] = L.item_count;
            L.item_count += nodes[v.id].count;
            continue;
        }
        const wide_node w = collapse(v.id);
        L.wide_index[v.id] = (int32_t)L.wide.size();
        L.wide.push_back(w);
        L.wide_of.push_back(v.id);
        const uint32_t pending = v.pending + (w.count - 1);
        L.stack_need = std::max(L.stack_need, pending);
        for (int k = (int)w.count - 1; k >= 0; k--) stack.push_back({ (uint32_t)w.child[k], pending });
    }
    return L;
}

// Writes the 64-byte nodes of a collapsed tree; `reference_to(sah id)` names a child (inner index or leaf reference).
template <class ReferenceTo>
void write_wide_nodes(const std::vector<sah_node>& nodes, const wide_layout& L, flat_bvh_node* out_nodes, ReferenceTo reference_to) {
    for (size_t wi = 0; wi < L.wide.size(); wi++) {
        const wide_node& w = L.wide[wi];
        const aabb& box = nodes[L.wide_of[wi]].box;
        flat_bvh_node& fn = out_nodes[wi];
        uint32_t biased[3];
        uint8_t qlo[3][4] = {}, qhi[3][4] = {};
        for (int a = 0; a < 3; a++) {
            float lo[4], hi[4];
            for (uint32_t k = 0; k < w.count; k++) { lo[k] = nodes[w.child[k]].box.minimum.v[a]; hi[k] = nodes[w.child[k]].box.maximum.v[a]; }
            quantise_axis(box.minimum.v[a], box.maximum.v[a], lo, hi, w.count, fn.origin[a], biased[a], qlo[a], qhi[a]);
        }
        fn.step_x = scale_from_biased(biased[0]);
        fn.step_y = scale_from_biased(biased[1]);
        fn.step_z = scale_from_biased(biased[2]);
        for (uint32_t k = 0; k < 4; k++) fn.child[k] = k < w.count ? reference_to(w.child[k]) : FLAT_BVH_EMPTY_CHILD;
        auto pack = [](const uint8_t* q) { return (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24); };
        fn.qlo[0] = pack(qlo[0]); fn.qlo[1] = pack(qlo[1]); fn.qlo[2] = pack(qlo[2]);
        fn.qhi_x = pack(qhi[0]); fn.qhi_y = pack(qhi[1]); fn.qhi_z = pack(qhi[2]);
    }
}

}  // namespace

std::vector<uint8_t> bvh::flatten(const std::vector<triangle>& only_tris, const std::vector<int32_t>& ranks, unsigned threads,
                                  const std::vector<flat_bvh_sphere>& spheres) {
    if (only_tris.empty() && spheres.empty()) throw bvh_build_error("flatten: no triangles");
    if (only_tris.size() + spheres.size() >= (1u << 29)) throw bvh_build_error("flatten: too many triangles for the leaf reference encoding");
    if (!ranks.empty() && ranks.size() != only_tris.size()) throw bvh_build_error("flatten: ranks/triangles size mismatch");
    for (const flat_bvh_sphere& s : spheres)
        if (!(s.radius > 0.0f) || !std::isfinite(s.radius) || !std::isfinite(s.center[0]) || !std::isfinite(s.center[1]) || !std::isfinite(s.center[2]))
            throw bvh_build_error("flatten: a sphere needs a finite centre and a positive finite radius");
    // the tree is built over the triangles and, behind them, one box-spanning proxy per sphere
    std::vector<triangle> with_proxies;
    if (!spheres.empty()) {
        with_proxies = only_tris;
        for (const flat_bvh_sphere& s : spheres) with_proxies.push_back(sphere_proxy(s));
    }
    const std::vector<triangle>& tris = spheres.empty() ? only_tris : with_proxies;
    const size_t triangle_count = only_tris.size();

    std::vector<int32_t> order;
    const std::vector<sah_node> nodes = build_sah(tris, order, 4, threads);
    const wide_layout L = collapse_tree(nodes);
    if (L.stack_need > FLAT_BVH_MAX_DEPTH) throw bvh_build_error("flatten: tree needs a deeper traversal stack than the kernels have");

    const size_t nodes_bytes = L.wide.size() * sizeof(flat_bvh_node);
    const size_t tris_bytes = tris.size() * sizeof(flat_bvh_triangle);
    std::vector<uint8_t> blob(sizeof(flat_bvh_header) + nodes_bytes + tris_bytes, 0);
    auto* header = reinterpret_cast<flat_bvh_header*>(blob.data());
    auto* out_nodes = reinterpret_cast<flat_bvh_node*>(blob.data() + sizeof(flat_bvh_header));
    auto* out_tris = reinterpret_cast<flat_bvh_triangle*>(blob.data() + sizeof(flat_bvh_header) + nodes_bytes);

    auto reference_to = [&](int32_t id) -> int32_t {
        const sah_node& nd = nodes[id];
        if (nd.left >= 0) return L.wide_index[id];
        return ~(int32_t)((L.leaf_first[id] << 2) | (nd.count - 1));
    };

    for (size_t id = 0; id < nodes.size(); id++) {
        const sah_node& nd = nodes[id];
        if (nd.left >= 0) continue;
        // leaves of collapsed-away inner nodes are still reached through their wide ancestor
        for (uint32_t k = 0; k < nd.count; k++) {
            const int32_t prim = order[nd.first + k];
            if ((size_t)prim < triangle_count) fill_flat_triangle(out_tris[L.leaf_first[id] + k], tris[prim], prim, ranks.empty() ? prim : ranks[prim]);
            else fill_flat_sphere(out_tris[L.leaf_first[id] + k], spheres[(size_t)prim - triangle_count], (int32_t)((size_t)prim - triangle_count), prim);
        }
    }
    write_wide_nodes(nodes, L, out_nodes, reference_to);

    header->magic = FLAT_BVH_MAGIC;
    header->version = FLAT_BVH_VERSION;
    header->node_count = (uint32_t)L.wide.size();
    header->triangle_count = (uint32_t)tris.size();
    header->nodes_offset = sizeof(flat_bvh_header);
    header->triangles_offset = sizeof(flat_bvh_header) + nodes_bytes;
    header->root = reference_to(0);
    header->max_depth = L.stack_need;
    for (int a = 0; a < 3; a++) { header->scene_min[a] = nodes[0].box.minimum.v[a]; header->scene_max[a] = nodes[0].box.maximum.v[a]; }
    return blob;
}

// ---- top level over instances of one flattened tree ----------------------------------------------------------
namespace {
// world-to-object of an affine object-to-world matrix (rows of 4: 3x3 part, translation), in double
bool invert_affine(const float* m, double inv[12]) {
    const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return false;
    const double r = 1.0 / det;
    inv[0] = (e * i - f * h) * r; inv[1] = (c * h - b * i) * r; inv[2] = (b * f - c * e) * r;
    inv[4] = (f * g - d * i) * r; inv[5] = (a * i - c * g) * r; inv[6] = (c * d - a * f) * r;
    inv[8] = (d * h - e * g) * r; inv[9] = (b * g - a * h) * r; inv[10] = (a * e - b * d) * r;
    for (int row = 0; row < 3; row++)
        inv[row * 4 + 3] = -(inv[row * 4] * (double)m[3] + inv[row * 4 + 1] * (double)m[7] + inv[row * 4 + 2] * (double)m[11]);
    for (int k = 0; k < 12; k++) if (!std::isfinite(inv[k])) return false;
    return true;
}
}  // namespace

std::vector<uint8_t> bvh::flatten_objects(const std::vector<triangle>& tris, const std::vector<int32_t>& ranks, const std::vector<uint32_t>& object_first,
                                          std::vector<flat_bvh_object>& objects, unsigned threads) {
    if (object_first.size() < 2 || object_first.front() != 0 || object_first.back() != tris.size()) throw bvh_build_error("flatten_objects: the offsets must run from 0 to the triangle count");
    if (tris.size() >= (1u << 28)) throw bvh_build_error("flatten_objects: too many triangles to be instanced");
    if (!ranks.empty() && ranks.size() != tris.size()) throw bvh_build_error("flatten_objects: ranks/triangles size mismatch");
    const size_t count = object_first.size() - 1;
    objects.assign(count, flat_bvh_object{});
    std::vector<flat_bvh_node> all_nodes;
    std::vector<flat_bvh_triangle> all_tris;
    all_tris.reserve(tris.size());
    aabb whole;
    uint32_t deepest = 0;
    for (size_t k = 0; k < count; k++) {
        const uint32_t first = object_first[k], last = object_first[k + 1];
        if (last <= first) throw bvh_build_error("flatten_objects: an object without triangles");
        const std::vector<triangle> part(tris.begin() + first, tris.begin() + last);
        std::vector<int32_t> part_ranks;
        if (!ranks.empty()) part_ranks.assign(ranks.begin() + first, ranks.begin() + last);
        const std::vector<uint8_t> blob = flatten(part, part_ranks, threads);
        const auto* h = reinterpret_cast<const flat_bvh_header*>(blob.data());
        const auto* nodes = reinterpret_cast<const flat_bvh_node*>(blob.data() + h->nodes_offset);
        const auto* records = reinterpret_cast<const flat_bvh_triangle*>(blob.data() + h->triangles_offset);
        const int32_t node_base = (int32_t)all_nodes.size();
        const uint32_t tri_base = (uint32_t)all_tris.size();
        // references move with the arrays: inner indices by the nodes before, leaf references by the triangles before
        auto moved = [&](int32_t ref) -> int32_t {
            if (ref == FLAT_BVH_EMPTY_CHILD) return ref;
            if (ref >= 0) return ref + node_base;
            const uint32_t r = (uint32_t)~ref;
            return ~(int32_t)((((r >> 2) + tri_base) << 2) | (r & 3u));
        };
        for (uint32_t i = 0; i < h->node_count; i++) {
            flat_bvh_node nd = nodes[i];
            for (int32_t& c : nd.child) c = moved(c);
            all_nodes.push_back(nd);
        }
        for (uint32_t i = 0; i < h->triangle_count; i++) {
            flat_bvh_triangle t = records[i];
            t.primitive_id += (int32_t)first;                       // global primitive ids
            if (ranks.empty()) t.rank += (int32_t)first;            // rank = primitive index
            all_tris.push_back(t);
        }
        flat_bvh_object& o = objects[k];
        o.root = moved(h->root);
        o.max_depth = h->max_depth;
        for (int a = 0; a < 3; a++) { o.box_min[a] = h->scene_min[a]; o.box_max[a] = h->scene_max[a]; }
        whole.union_with(aabb(vec3(h->scene_min[0], h->scene_min[1], h->scene_min[2]), vec3(h->scene_max[0], h->scene_max[1], h->scene_max[2])));
        deepest = std::max(deepest, h->max_depth);
    }
    const size_t nodes_bytes = all_nodes.size() * sizeof(flat_bvh_node);
    std::vector<uint8_t> blob(sizeof(flat_bvh_header) + nodes_bytes + all_tris.size() * sizeof(flat_bvh_triangle), 0);
    auto* header = reinterpret_cast<flat_bvh_header*>(blob.data());
    if (!all_nodes.empty()) std::memcpy(blob.data() + sizeof(flat_bvh_header), all_nodes.data(), nodes_bytes);
    std::memcpy(blob.data() + sizeof(flat_bvh_header) + nodes_bytes, all_tris.data(), all_tris.size() * sizeof(flat_bvh_triangle));
    header->magic = FLAT_BVH_MAGIC;
    header->version = FLAT_BVH_VERSION;
    header->node_count = (uint32_t)all_nodes.size();
    header->triangle_count = (uint32_t)all_tris.size();
    header->nodes_offset = sizeof(flat_bvh_header);
    header->triangles_offset = sizeof(flat_bvh_header) + nodes_bytes;
    header->root = objects[0].root;
    header->max_depth = deepest;
    for (int a = 0; a < 3; a++) { header->scene_min[a] = whole.minimum.v[a]; header->scene_max[a] = whole.maximum.v[a]; }
    return blob;
}

std::vector<uint8_t> bvh::flatten_instances(const std::vector<uint8_t>& object_blob, const std::vector<float>& object_to_world, unsigned threads) {
    if (object_blob.size() < sizeof(flat_bvh_header)) throw bvh_build_error("flatten_instances: no object tree");
    const auto* object_header = reinterpret_cast<const flat_bvh_header*>(object_blob.data());
    if (object_header->magic != FLAT_BVH_MAGIC || object_header->version != FLAT_BVH_VERSION) throw bvh_build_error("flatten_instances: not a flattened tree");
    if (object_header->triangle_count >= (1u << 28)) throw bvh_build_error("flatten_instances: object tree too large to be instanced");
    flat_bvh_object whole;
    whole.root = object_header->root;
    whole.max_depth = object_header->max_depth;
    for (int a = 0; a < 3; a++) { whole.box_min[a] = object_header->scene_min[a]; whole.box_max[a] = object_header->scene_max[a]; }
    return flatten_instances(std::vector<flat_bvh_object>{ whole }, std::vector<uint32_t>(object_to_world.size() / 12, 0u), object_to_world, threads);
}

std::vector<uint8_t> bvh::flatten_instances(const std::vector<flat_bvh_object>& objects, const std::vector<uint32_t>& object_of_copy,
                                            const std::vector<float>& object_to_world, unsigned threads) {
    if (objects.empty()) throw bvh_build_error("flatten_instances: no object");
    if (object_to_world.empty() || object_to_world.size() % 12 != 0) throw bvh_build_error("flatten_instances: transforms are 12 floats each");
    const size_t count = object_to_world.size() / 12;
    if (object_of_copy.size() != count) throw bvh_build_error("flatten_instances: one object index per transform");
    if (count >= FLAT_INSTANCE_LIMIT) throw bvh_build_error("flatten_instances: too many instances for the leaf reference encoding");

    // World box of each instance: the eight corners of the object's box through the matrix (double), rounded
    // outwards and padded by 2^-18 of the box's size and distance from the origin -- room for the rounding of the
    // object-space ray (a few 2^-24 of the same magnitudes) that the traversal below the instance sees.
    std::vector<flat_bvh_instance> records(count);
    std::vector<triangle> proxies(count);  // a "triangle" spanning the box: same bounds, centroid near the middle
    uint32_t object_depth = 0;
    for (size_t k = 0; k < count; k++) {
        if (object_of_copy[k] >= objects.size()) throw bvh_build_error("flatten_instances: a copy names an object that does not exist");
        const flat_bvh_object& object = objects[object_of_copy[k]];
        object_depth = std::max(object_depth, object.max_depth);
        const float* m = &object_to_world[k * 12];
        for (int j = 0; j < 12; j++) if (!std::isfinite(m[j])) throw bvh_build_error("flatten_instances: non-finite transform");
        double inv[12];
        if (!invert_affine(m, inv)) throw bvh_build_error("flatten_instances: singular transform");
        double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
        for (int corner = 0; corner < 8; corner++) {
            const double p[3] = { (corner & 1) ? object.box_max[0] : object.box_min[0],
                                  (corner & 2) ? object.box_max[1] : object.box_min[1],
                                  (corner & 4) ? object.box_max[2] : object.box_min[2] };
            for (int a = 0; a < 3; a++) {
                const double w = (double)m[a * 4] * p[0] + (double)m[a * 4 + 1] * p[1] + (double)m[a * 4 + 2] * p[2] + (double)m[a * 4 + 3];
                lo[a] = std::min(lo[a], w);
                hi[a] = std::max(hi[a], w);
            }
        }
        double magnitude = 0.0;
        for (int a = 0; a < 3; a++) magnitude = std::max({ magnitude, std::fabs(lo[a]), std::fabs(hi[a]), hi[a] - lo[a] });
        const double pad = magnitude * 3.814697265625e-06 + 1e-30;  // 2^-18
        vec3 bmin, bmax;
        for (int a = 0; a < 3; a++) {
            bmin.v[a] = std::nextafter((float)(lo[a] - pad), -std::numeric_limits<float>::infinity());
            bmax.v[a] = std::nextafter((float)(hi[a] + pad), std::numeric_limits<float>::infinity());
            if (!std::isfinite(bmin.v[a]) || !std::isfinite(bmax.v[a])) throw bvh_build_error("flatten_instances: instance out of the fp32 range");
        }
        proxies[k] = triangle(bmin, bmax, (bmin + bmax) / 2.f);
        flat_bvh_instance& r = records[k];
        for (int j = 0; j < 12; j++) r.world_to_object[j] = (float)inv[j];
        r.object_root = object.root;
        r.pad[0] = r.pad[1] = r.pad[2] = 0;
    }

    std::vector<int32_t> order;
    const std::vector<sah_node> nodes = build_sah(proxies, order, 1, threads);
    const wide_layout L = collapse_tree(nodes);
    // one entry marks the way back out of an instance; the object's own stack sits on top of it
    const uint32_t stack_need = L.stack_need + 1u + object_depth;
    if (stack_need > FLAT_BVH_MAX_DEPTH) throw bvh_build_error("flatten_instances: the two trees need a deeper traversal stack than the kernels have");

    const size_t nodes_bytes = L.wide.size() * sizeof(flat_bvh_node);
    std::vector<uint8_t> blob(sizeof(flat_bvh_header) + nodes_bytes + count * sizeof(flat_bvh_instance), 0);
    auto* header = reinterpret_cast<flat_bvh_header*>(blob.data());
    auto* out_nodes = reinterpret_cast<flat_bvh_node*>(blob.data() + sizeof(flat_bvh_header));
    auto reference_to = [&](int32_t id) -> int32_t {
        const sah_node& nd = nodes[id];
        if (nd.left >= 0) return L.wide_index[id];
        return ~(int32_t)(FLAT_INSTANCE_TAG | (uint32_t)order[nd.first]);  // one instance per leaf, records stay in the caller's order
    };
    write_wide_nodes(nodes, L, out_nodes, reference_to);
    std::memcpy(blob.data() + sizeof(flat_bvh_header) + nodes_bytes, records.data(), count * sizeof(flat_bvh_instance));

    header->magic = FLAT_INSTANCES_MAGIC;
    header->version = FLAT_BVH_VERSION;
    header->node_count = (uint32_t)L.wide.size();
    header->triangle_count = (uint32_t)count;
    header->nodes_offset = sizeof(flat_bvh_header);
    header->triangles_offset = sizeof(flat_bvh_header) + nodes_bytes;
    header->root = reference_to(0);
    header->max_depth = stack_need;
    for (int a = 0; a < 3; a++) { header->scene_min[a] = nodes[0].box.minimum.v[a]; header->scene_max[a] = nodes[0].box.maximum.v[a]; }
    return blob;
}
