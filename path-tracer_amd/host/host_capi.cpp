// C entry points over the host-side scene-preparation code (bvh, camera) so that the test-suite and
// the benchmark driver can call it through ctypes.  Declared in include/ptr_host.h.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "bvh.hpp"
#include "camera.hpp"

namespace {
thread_local std::string g_last_error;

std::vector<triangle> make_triangles(const float* positions, int64_t n) {
    std::vector<triangle> tris;
    tris.reserve((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        const float* p = positions + i * 9;
        tris.emplace_back(vec3{ p[0], p[1], p[2] }, vec3{ p[3], p[4], p[5] }, vec3{ p[6], p[7], p[8] });
    }
    return tris;
}
}  // namespace

extern "C" {

const char* ptrh_last_error(void) { return g_last_error.c_str(); }

void ptrh_sizes(uint32_t out[4]) {
    out[0] = (uint32_t)sizeof(packed_bvh_node);
    out[1] = (uint32_t)sizeof(vec3);
    out[2] = (uint32_t)sizeof(camera);
    out[3] = (uint32_t)sizeof(triangle);
}

int64_t ptrh_bvh_build(const float* positions, int64_t n, int32_t mode, void* nodes_out) {
    try {
        if (n <= 0) { g_last_error = "ptrh_bvh_build: n must be positive"; return -1; }
        std::vector<triangle> tris = make_triangles(positions, n);
        std::vector<packed_bvh_node> packed;
        bvh builder(tris, packed, mode == 0 ? bvh::build_mode::reference_order : bvh::build_mode::sah_partitioned);
        std::memcpy(nodes_out, packed.data(), packed.size() * sizeof(packed_bvh_node));
        return (int64_t)packed.size();
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return -1;
    }
}

int32_t ptrh_bvh_visit_ranks(const void* nodes, int64_t node_count, int64_t primitive_count, int32_t* ranks_out) {
    try {
        std::vector<packed_bvh_node> packed((size_t)node_count);
        std::memcpy(packed.data(), nodes, (size_t)node_count * sizeof(packed_bvh_node));
        const std::vector<int32_t> r = bvh::visit_ranks(packed, (size_t)primitive_count);
        std::memcpy(ranks_out, r.data(), r.size() * sizeof(int32_t));
        return 0;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return -1;
    }
}

int32_t ptrh_bvh_flatten(const float* positions, int64_t n, const int32_t* ranks, uint32_t threads, void** blob_out,
                         uint64_t* size_out) {
    try {
        if (n <= 0) { g_last_error = "ptrh_bvh_flatten: n must be positive"; return -1; }
        const std::vector<triangle> tris = make_triangles(positions, n);
        std::vector<int32_t> r;
        if (ranks) r.assign(ranks, ranks + n);
        const std::vector<uint8_t> blob = bvh::flatten(tris, r, threads);
        void* mem = std::malloc(blob.size());
        if (!mem) { g_last_error = "ptrh_bvh_flatten: out of memory"; return -1; }
        std::memcpy(mem, blob.data(), blob.size());
        *blob_out = mem;
        *size_out = blob.size();
        return 0;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return -1;
    }
}

void ptrh_free(void* p) { std::free(p); }

void ptrh_camera(const float pos[3], const float target[3], float v_fov, float aspect, float aperture, float focus_dist,
                 const float* ops, int32_t n_ops, void* out) {
    camera cam(point3{ pos[0], pos[1], pos[2] }, point3{ target[0], target[1], target[2] }, v_fov, aspect, aperture,
               focus_dist);
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
