// C entry points over the host-side scene-preparation code (bvh, camera) so that the test-suite and
// the benchmark driver can call it through ctypes.  Declared in include/ptr_host.h.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "bvh.hpp"
#include "camera.hpp"
#include "gltf.hpp"
#include "image.hpp"
#include "json.hpp"
#include "scene_source.hpp"

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

int32_t ptrh_bvh_flatten_spheres(const float* positions, int64_t n, const int32_t* ranks, const float* spheres, int64_t sphere_count, uint32_t threads,
                                 void** blob_out, uint64_t* size_out) {
    try {
        if (n < 0 || sphere_count < 0 || n + sphere_count <= 0 || (sphere_count && !spheres)) { g_last_error = "ptrh_bvh_flatten_spheres: no primitives"; return -1; }
        const std::vector<triangle> tris = n ? make_triangles(positions, n) : std::vector<triangle>();
        std::vector<int32_t> r;
        if (ranks) r.assign(ranks, ranks + n);
        std::vector<flat_bvh_sphere> s((size_t)sphere_count);
        if (sphere_count) std::memcpy(s.data(), spheres, (size_t)sphere_count * sizeof(flat_bvh_sphere));
        const std::vector<uint8_t> blob = bvh::flatten(tris, r, threads, s);
        void* mem = std::malloc(blob.size());
        if (!mem) { g_last_error = "ptrh_bvh_flatten_spheres: out of memory"; return -1; }
        std::memcpy(mem, blob.data(), blob.size());
        *blob_out = mem;
        *size_out = blob.size();
        return 0;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return -1;
    }
}

int32_t ptrh_random_spheres(uint32_t seed, float* out9, uint32_t capacity, uint32_t* count_out) {
    try {
        if (!count_out || (capacity && !out9)) { g_last_error = "ptrh_random_spheres: null argument"; return -1; }
        const std::vector<scene_source::source_sphere> all = scene_source::random_spheres(seed);
        *count_out = (uint32_t)all.size();
        for (uint32_t i = 0; i < capacity && i < all.size(); i++) {
            const auto& s = all[i];
            const float rec[9] = { s.center[0], s.center[1], s.center[2], s.radius, s.material.base_color[0], s.material.base_color[1], s.material.base_color[2],
                                   s.material.metalness, s.material.roughness };
            std::memcpy(out9 + 9 * (size_t)i, rec, sizeof rec);
        }
        return 0;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return -1;
    }
}

int32_t ptrh_bvh_flatten_instances(const void* object_blob, uint64_t object_bytes, const float* object_to_world, int64_t instance_count,
                                   uint32_t threads, void** blob_out, uint64_t* size_out) {
    try {
        if (!object_blob || !object_to_world || !blob_out || !size_out) { g_last_error = "ptrh_bvh_flatten_instances: null argument"; return -1; }
        if (instance_count <= 0) { g_last_error = "ptrh_bvh_flatten_instances: instance_count must be positive"; return -1; }
        const std::vector<uint8_t> object((const uint8_t*)object_blob, (const uint8_t*)object_blob + object_bytes);
        const std::vector<float> transforms(object_to_world, object_to_world + instance_count * 12);
        const std::vector<uint8_t> blob = bvh::flatten_instances(object, transforms, threads);
        void* mem = std::malloc(blob.size());
        if (!mem) { g_last_error = "ptrh_bvh_flatten_instances: out of memory"; return -1; }
        std::memcpy(mem, blob.data(), blob.size());
        *blob_out = mem;
        *size_out = blob.size();
        return 0;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return -1;
    }
}

int32_t ptrh_bvh_flatten_objects(const float* positions, int64_t n, const int32_t* ranks, const uint32_t* object_first, int64_t object_count,
                                 uint32_t threads, void** blob_out, uint64_t* size_out, void* objects_out) {
    try {
        if (!positions || !object_first || !blob_out || !size_out || !objects_out) { g_last_error = "ptrh_bvh_flatten_objects: null argument"; return -1; }
        if (n <= 0 || object_count <= 0) { g_last_error = "ptrh_bvh_flatten_objects: counts must be positive"; return -1; }
        const std::vector<triangle> tris = make_triangles(positions, n);
        std::vector<int32_t> r;
        if (ranks) r.assign(ranks, ranks + n);
        const std::vector<uint32_t> first(object_first, object_first + object_count + 1);
        std::vector<flat_bvh_object> objects;
        const std::vector<uint8_t> blob = bvh::flatten_objects(tris, r, first, objects, threads);
        void* mem = std::malloc(blob.size());
        if (!mem) { g_last_error = "ptrh_bvh_flatten_objects: out of memory"; return -1; }
        std::memcpy(mem, blob.data(), blob.size());
        std::memcpy(objects_out, objects.data(), objects.size() * sizeof(flat_bvh_object));
        *blob_out = mem;
        *size_out = blob.size();
        return 0;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return -1;
    }
}

int32_t ptrh_bvh_flatten_placements(const void* objects, int64_t object_count, const uint32_t* object_of_copy, const float* object_to_world,
                                    int64_t copy_count, uint32_t threads, void** blob_out, uint64_t* size_out) {
    try {
        if (!objects || !object_of_copy || !object_to_world || !blob_out || !size_out) { g_last_error = "ptrh_bvh_flatten_placements: null argument"; return -1; }
        if (object_count <= 0 || copy_count <= 0) { g_last_error = "ptrh_bvh_flatten_placements: counts must be positive"; return -1; }
        std::vector<flat_bvh_object> table((size_t)object_count);
        std::memcpy(table.data(), objects, table.size() * sizeof(flat_bvh_object));
        const std::vector<uint32_t> which(object_of_copy, object_of_copy + copy_count);
        const std::vector<float> transforms(object_to_world, object_to_world + copy_count * 12);
        const std::vector<uint8_t> blob = bvh::flatten_instances(table, which, transforms, threads);
        void* mem = std::malloc(blob.size());
        if (!mem) { g_last_error = "ptrh_bvh_flatten_placements: out of memory"; return -1; }
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

// ---- glTF -> flat scene arrays ---------------------------------------------------------------------------
int32_t ptrh_gltf_load(const char* path, int32_t apply_transforms, int32_t wide_indices, void** scene_out) {
    try {
        if (!path || !scene_out) { g_last_error = "ptrh_gltf_load: null argument"; return -1; }
        const gltf model{ std::string(path) };
        *scene_out = new scene_source(apply_transforms == 2 ? model.flatten_instanced(wide_indices) : model.flatten(apply_transforms != 0, wide_indices));
        return 0;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return -1;
    }
}

int32_t ptrh_scene_load(const char* path, void** scene_out) {
    try {
        if (!path || !scene_out) { g_last_error = "ptrh_scene_load: null argument"; return -1; }
        *scene_out = new scene_source(scene_source::load(path));
        return 0;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return -1;
    }
}

int32_t ptrh_scene_counts(const void* scene, uint64_t counts[5]) {
    if (!scene || !counts) { g_last_error = "ptrh_scene_counts: null argument"; return -1; }
    const scene_source* s = (const scene_source*)scene;
    counts[0] = s->indices.size(); counts[1] = s->positions.size() / 3; counts[2] = s->materials.size(); counts[3] = s->textures.size();
    counts[4] = s->index_words;
    return 0;
}

int32_t ptrh_scene_arrays(const void* scene, uint32_t* indices, float* positions, float* normals, float* uvs, void* materials) {
    if (!scene) { g_last_error = "ptrh_scene_arrays: null scene"; return -1; }
    const scene_source* s = (const scene_source*)scene;
    static_assert(sizeof(scene_source::source_material) == 32, "material record handed to C callers is 32 bytes");
    if (indices) std::memcpy(indices, s->indices.data(), s->indices.size() * 4);
    if (positions) std::memcpy(positions, s->positions.data(), s->positions.size() * 4);
    if (normals) std::memcpy(normals, s->normals.data(), s->normals.size() * 4);
    if (uvs) std::memcpy(uvs, s->uvs.data(), s->uvs.size() * 4);
    if (materials) std::memcpy(materials, s->materials.data(), s->materials.size() * sizeof(scene_source::source_material));
    return 0;
}

int32_t ptrh_scene_texture(const void* scene, uint32_t index, uint32_t* width, uint32_t* height, uint8_t* rgba) {
    if (!scene) { g_last_error = "ptrh_scene_texture: null scene"; return -1; }
    const scene_source* s = (const scene_source*)scene;
    if (index >= s->textures.size()) { g_last_error = "ptrh_scene_texture: index out of range"; return -1; }
    const auto& t = s->textures[index];
    if (width) *width = t.width;
    if (height) *height = t.height;
    if (rgba) std::memcpy(rgba, t.rgba.data(), t.rgba.size());
    return 0;
}

int32_t ptrh_scene_placements(const void* scene, uint64_t counts[2], uint32_t* object_first, uint32_t* object_of_copy, float* object_to_world) {
    if (!scene || !counts) { g_last_error = "ptrh_scene_placements: null argument"; return -1; }
    const scene_source* s = (const scene_source*)scene;
    counts[0] = s->object_first.empty() ? 0 : s->object_first.size() - 1;
    counts[1] = s->placements.size();
    if (object_first) std::memcpy(object_first, s->object_first.data(), s->object_first.size() * 4);
    for (size_t i = 0; i < s->placements.size(); i++) {
        if (object_of_copy) object_of_copy[i] = s->placements[i].object;
        if (object_to_world) std::memcpy(object_to_world + i * 12, s->placements[i].object_to_world, sizeof(float) * 12);
    }
    return 0;
}

void ptrh_scene_free(void* scene) { delete (scene_source*)scene; }

int32_t ptrh_image_decode(const uint8_t* bytes, uint64_t size, int32_t* width, int32_t* height, int32_t* channels, uint8_t** rgba_out) {
    try {
        if (!bytes || !width || !height || !channels || !rgba_out) { g_last_error = "ptrh_image_decode: null argument"; return -1; }
        const raw_image im = decode_image(bytes, (size_t)size);
        *width = im.width; *height = im.height; *channels = im.channels;
        *rgba_out = (uint8_t*)std::malloc(im.data.size() ? im.data.size() : 1);
        if (!*rgba_out) { g_last_error = "ptrh_image_decode: out of memory"; return -1; }
        std::memcpy(*rgba_out, im.data.data(), im.data.size());
        return 0;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return -1;
    }
}

int32_t ptrh_image_encode_png(const uint8_t* rgb, uint32_t width, uint32_t height, uint8_t** file_out, uint64_t* size_out) {
    try {
        if (!rgb || !file_out || !size_out) { g_last_error = "ptrh_image_encode_png: null argument"; return -1; }
        const std::vector<uint8_t> file = encode_png_rgb8(rgb, width, height);
        *file_out = (uint8_t*)std::malloc(file.size());
        if (!*file_out) { g_last_error = "ptrh_image_encode_png: out of memory"; return -1; }
        std::memcpy(*file_out, file.data(), file.size());
        *size_out = file.size();
        return 0;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return -1;
    }
}

int32_t ptrh_json_dump(const char* text, uint64_t size, char** out) {
    try {
        if (!text || !out) { g_last_error = "ptrh_json_dump: null argument"; return -1; }
        const std::string d = json::parse(std::string(text, (size_t)size)).dump();
        *out = (char*)std::malloc(d.size() + 1);
        if (!*out) { g_last_error = "ptrh_json_dump: out of memory"; return -1; }
        std::memcpy(*out, d.c_str(), d.size() + 1);
        return 0;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return -1;
    }
}

}  // extern "C"
