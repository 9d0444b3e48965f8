// BVH builder: keeps the reference's public API (include/bvh.hpp:9-60 in the reference:
// `bvh(std::vector<triangle>&, std::vector<packed_bvh_node>&)`, `packed_bvh_node`, `exporter()`)
// and adds what the MI355X kernels need:
//   * build_mode::reference_order  -- the reference's top-down bucketed build (src/bvh.cpp:150-262),
//     reproduced bit for bit: same threaded 32-byte node array, same depth-first numbering.
//   * build_mode::sah_partitioned  -- a real binned-SAH build (primitives ARE partitioned).
//   * flatten()                    -- the GPU layout: 64-byte, cache-line-aligned nodes of a 4-wide tree
//     with quantised child boxes, and 64-byte pre-gathered triangle records in leaf order.
#ifndef PTR_BVH_HPP_
#define PTR_BVH_HPP_

#include <cstdint>
#include <stdexcept>
#include <vector>

#include "aabb.hpp"
#include "triangle.hpp"

// Threaded (skip-link) node, the layout the reference shader walks (compute.comp:60-65):
// depth-first order, left child = id + 1, next_id = where to go on a miss (-1 = done).
struct packed_bvh_node {
    float min[3];
    int32_t next_id = -1;
    float max[3];
    int32_t primitive_id = -1;
};
static_assert(sizeof(packed_bvh_node) == 32, "threaded node is 32 bytes");

// ---- GPU layout -------------------------------------------------------------------------------
// One 64-byte line per inner node of a 4-wide tree.  The node stores an origin just below its own box
// (fp32) and, per axis, a power-of-two step; the boxes of its up to four children are 8-bit offsets in
// those steps, rounded outwards with a margin of 2^-20 of the node extent, so a decoded box always
// contains the true one with room for the kernels' rounding:
//     decoded = origin + q * step          (q * step is exact, one rounding in the add)
// Byte k of each q word belongs to child k.  A child reference >= 0 is an inner-node index; < 0 is a leaf:
// ~ref = (first_triangle << 2) | (triangle_count - 1), 1..4 triangles per leaf; unused slots (always the
// last ones, children 0 and 1 always exist) hold FLAT_BVH_EMPTY_CHILD.
// Four 16-byte quarters = four loads per lane for four boxes.
struct flat_bvh_node {
    float origin[3]; float step_x;
    int32_t child[4];
    uint32_t qlo[3]; uint32_t qhi_x;   // lower bounds x, y, z and upper bound x of the four children
    uint32_t qhi_y, qhi_z; float step_y, step_z;
};
static_assert(sizeof(flat_bvh_node) == 64, "flat node is one 64-byte line");

// One 64-byte line per triangle, in leaf order.  e1, e2 and n are the ray-independent terms of the
// reference's intersection test (ray.h:142-146): e1 = p2 - p1, e2 = p3 - p1, n = cross(e1, e2),
// computed here in fp32 with the shader's operation order so the kernel's u, v, t keep their bits.
// `rank` is the triangle's position in the reference traversal order (tie-break on equal t:
// the reference keeps the later-visited triangle, ray.h:153).
//
// A record can also hold a sphere -- the primitive of the reference's hit_sphere (shaders/include/ray.h:105-135), whose call
// site in hit_node is commented out (ray.h:173-177): kind = FLAT_BVH_KIND_SPHERE, p0 = centre, e1[0] = radius, primitive_id =
// FLAT_BVH_SPHERE_TAG | index into the sphere array bound beside the layout; e2 and n are unused.
struct flat_bvh_triangle {
    float p0[3]; int32_t rank;
    float e1[3]; int32_t primitive_id;
    float e2[3]; int32_t kind;
    float n[3];  int32_t pad1;
};
constexpr int32_t FLAT_BVH_KIND_TRIANGLE = 0, FLAT_BVH_KIND_SPHERE = 1;
constexpr int32_t FLAT_BVH_SPHERE_TAG = 0x40000000;   // triangle ids stay below 2^29
struct flat_bvh_sphere { float center[3]; float radius; };
static_assert(sizeof(flat_bvh_triangle) == 64, "flat triangle is one 64-byte line");

struct flat_bvh_header {
    uint32_t magic;            // 'PTRF'
    uint32_t version;
    uint32_t node_count;
    uint32_t triangle_count;
    uint64_t nodes_offset;     // byte offsets from the start of the blob, 64-byte aligned
    uint64_t triangles_offset;
    int32_t root;              // root reference (a leaf reference when the scene has <= 4 triangles)
    uint32_t max_depth;        // most stack entries a traversal can hold at once (bounds the kernels' stacks)
    float scene_min[3];
    float scene_max[3];
};
static_assert(sizeof(flat_bvh_header) == 64, "header is one 64-byte line");

constexpr uint32_t FLAT_BVH_MAGIC = 0x46525450u;  // "PTRF"
constexpr uint32_t FLAT_BVH_VERSION = 2;          // 4-wide nodes with quantised child boxes
constexpr uint32_t FLAT_BVH_MAX_DEPTH = 254;      // most stack entries a layout may ask for (the kernels size their spill area from the header)
constexpr int32_t FLAT_BVH_EMPTY_CHILD = INT32_MIN;

// ---- two levels: instances of one flattened tree ------------------------------------------------------------
// A second blob of the same shape (header | nodes | records) whose leaves name instances instead of triangles:
// ~ref = FLAT_INSTANCE_TAG | instance index, one instance per leaf.  Its boxes are in world space; below an instance
// the traversal goes on in the object's space with the ray taken through `world_to_object` (the direction is not
// renormalised, so a distance t means the same on both sides).  header.triangle_count holds the instance count,
// header.triangles_offset the offset of the records, header.max_depth the stack of both levels together.
struct flat_bvh_instance {
    float world_to_object[12];   // three rows of (3x3 part, translation)
    int32_t object_root;         // root reference of the instanced tree
    int32_t pad[3];
};
static_assert(sizeof(flat_bvh_instance) == 64, "instance record is one 64-byte line");
// One object inside a flattened layout that holds several (bvh::flatten_objects): where its tree starts and what it needs.
struct flat_bvh_object {
    int32_t root;
    uint32_t max_depth;
    float box_min[3];
    float box_max[3];
};
static_assert(sizeof(flat_bvh_object) == 32, "object entry is eight words");
constexpr uint32_t FLAT_INSTANCES_MAGIC = 0x49525450u;  // "PTRI"
constexpr uint32_t FLAT_INSTANCE_TAG = 0x40000000u;      // set in ~ref of an instance leaf; triangle leaves of an instanced tree stay below it
constexpr uint32_t FLAT_INSTANCE_LIMIT = 0x3ffffff0u;    // instance indices above are reserved for the kernels' markers

class bvh_build_error : public std::runtime_error {
public:
    using std::runtime_error::runtime_error;
};

class bvh {
public:
    enum class build_mode { reference_order, sah_partitioned };

    // Reference API.  Throws bvh_build_error where the reference would recurse without end
    // (all centroids of a >4-triangle range equal on the split axis, src/bvh.cpp:208-216).
    bvh(std::vector<triangle>& triangles, std::vector<packed_bvh_node>& packed_nodes);
    bvh(std::vector<triangle>& triangles, std::vector<packed_bvh_node>& packed_nodes, build_mode mode);

    // Graphviz dump of the threaded tree (the reference's debugging aid, src/bvh.cpp:91-99).
    void exporter();

    // Position of every primitive in the threaded tree's leaf visiting order.
    static std::vector<int32_t> visit_ranks(const std::vector<packed_bvh_node>& packed_nodes, size_t primitive_count);

    // GPU layout.  `ranks` (one per primitive) may be empty: then rank = primitive index.
    // Returns one contiguous blob: header | nodes | triangles.
    // `spheres`: further primitives of the same tree (sphere leaves); sphere s hits report primitive FLAT_BVH_SPHERE_TAG | s
    // and rank triangles.size() + s.
    static std::vector<uint8_t> flatten(const std::vector<triangle>& triangles, const std::vector<int32_t>& ranks,
                                        unsigned threads = 0, const std::vector<flat_bvh_sphere>& spheres = {});

    // Several objects in ONE layout: triangles [object_first[k], object_first[k + 1]) form object k, each with a tree of
    // its own; node and triangle records of all objects share the blob's two arrays and primitive ids stay global, so the
    // index / vertex / shading buffers need no per-object base.  The header names object 0's root, the union of the boxes
    // and the deepest stack; `objects` receives one entry per object.
    static std::vector<uint8_t> flatten_objects(const std::vector<triangle>& triangles, const std::vector<int32_t>& ranks,
                                                const std::vector<uint32_t>& object_first, std::vector<flat_bvh_object>& objects, unsigned threads = 0);
    // Top level over placed copies of those objects: copy i shows objects[object_of_copy[i]] through object_to_world[12 i ..].
    static std::vector<uint8_t> flatten_instances(const std::vector<flat_bvh_object>& objects, const std::vector<uint32_t>& object_of_copy,
                                                  const std::vector<float>& object_to_world, unsigned threads = 0);

    // Top level over `object_to_world.size() / 12` instances (rows of an affine object-to-world matrix each) of the
    // tree in `object_blob`.  Returns header | nodes | instance records.
    static std::vector<uint8_t> flatten_instances(const std::vector<uint8_t>& object_blob, const std::vector<float>& object_to_world,
                                                  unsigned threads = 0);

    std::vector<triangle>& triangles;

private:
    const std::vector<packed_bvh_node>* packed = nullptr;
};

#endif  // PTR_BVH_HPP_
