// TEST INFRASTRUCTURE ONLY -- never linked into, or called from, the product path.
//
// Thin C-ABI harness around the *unmodified* reference host sources
// (/root/reference/src/{bvh,vec3,camera}.cpp, compiled where they lie by oracle/Makefile
// into oracle/_ref/libptref.so).  It exists to (1) generate the golden fixtures under
// tests/golden/ (oracle/gen_golden.py) and (2) cross-check this repository's own host code
// (bvh / camera re-implementation) against the real reference when /root/reference is present.
//
// Reference entry points wrapped here:
//   bvh::bvh(std::vector<triangle>&, std::vector<packed_bvh_node>&)   src/bvh.cpp:48-86
//   camera::camera / set_aspect_ratio / move / rotate_y                 src/camera.cpp:6-57
//   struct layouts                                                      include/bvh.hpp:17-22, include/camera.hpp:6-39
#include <cstdint>
#include <cstring>
#include <vector>

#include "bvh.hpp"
#include "camera.hpp"

extern "C" {

// sizes the GPU ABI depends on: [packed_bvh_node, vec3, camera, triangle]
void ptref_sizes(uint32_t out[4]) {
    out[0] = (uint32_t)sizeof(packed_bvh_node);
    out[1] = (uint32_t)sizeof(vec3);
    out[2] = (uint32_t)sizeof(camera);
    out[3] = (uint32_t)sizeof(triangle);
}

// positions: n*9 floats (p1.xyz p2.xyz p3.xyz per triangle); nodes_out: (2n-1)*32 bytes.
// Returns the node count.
int64_t ptref_bvh_build(const float* positions, int64_t n, void* nodes_out) {
    std::vector<triangle> tris;
    tris.reserve((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        const float* p = positions + i * 9;
        tris.emplace_back(vec3{ p[0], p[1], p[2] }, vec3{ p[3], p[4], p[5] }, vec3{ p[6], p[7], p[8] });
    }
    std::vector<packed_bvh_node> packed;
    bvh builder(tris, packed);
    std::memcpy(nodes_out, packed.data(), packed.size() * sizeof(packed_bvh_node));
    return (int64_t)packed.size();
}

// ops: sequence of (opcode, a, b, c) quadruples applied after construction:
//   0 = set_aspect_ratio(a), 1 = move(vec3(a,b,c)), 2 = rotate_y(a)
// out: sizeof(camera) (=128) bytes, the struct exactly as scene::metadata embeds it.
void ptref_camera(const float pos[3], const float target[3], float v_fov, float aspect, float aperture,
                  float focus_dist, const float* ops, int32_t n_ops, void* out) {
    camera cam(point3{ pos[0], pos[1], pos[2] }, point3{ target[0], target[1], target[2] }, v_fov, aspect,
               aperture, focus_dist);
    for (int32_t i = 0; i < n_ops; i++) {
        const float* o = ops + i * 4;
        switch ((int)o[0]) {
        case 0: cam.set_aspect_ratio(o[1]); break;
        case 1: cam.move(vec3{ o[1], o[2], o[3] }); break;
        case 2: cam.rotate_y(o[1]); break;
        default: break;
        }
    }
    std::memcpy(out, &cam, sizeof(camera));
}

}  // extern "C"
