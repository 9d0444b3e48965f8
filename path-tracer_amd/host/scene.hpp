// Scene: owns the per-frame metadata block and the seven device buffers the kernels read, with the member
// names of the reference (include/scene.hpp:10-43).  The reference's constructor parses a hard-coded glTF
// file; here the flattened arrays come from a `scene_source` (a glTF loader or a .ptrscene file fills it),
// and the flattening that follows is the reference's (src/scene.cpp:84-204) with its buffer-sizing bugs fixed.
#ifndef PTR_SCENE_HPP_
#define PTR_SCENE_HPP_

#include <cstdint>
#include <string>
#include <vector>

#include "camera.hpp"
#include "material.hpp"
#include "scene_source.hpp"

class Buffer;
class Texture;
class Sampler;

class scene {
public:
    struct metadata {
        camera cam;

        uint32_t max_bounce = 3;
        uint32_t min_bounce = 1;

        uint32_t width;
        uint32_t height;

        uint32_t sample_index = 1;  // the reference starts at 0 and divides by it; 1 is the first valid sample

        uint32_t enable_dof = 0;
        uint32_t debug_bvh = 0;
        int32_t downscale_factor = 1;

        metadata(const camera& cam, uint32_t width, uint32_t height);
    };
    static_assert(sizeof(camera) == 128, "camera block");

    // flatten_bvh: also build the flattened layout the wavefront pipeline traverses -- on the host (binned SAH, the better
    // tree) or, with gpu_bvh, on the GPU from the uploaded buffers (ptr_bvh_build_flat: milliseconds instead of seconds).
    scene(const camera& cam, uint32_t width, uint32_t height, const scene_source& source, bool flatten_bvh = true, bool gpu_bvh = false);
    ~scene();

    // Instancing (an extension; the reference writes repeated objects out copy by copy): renders `object_to_world.size() / 12`
    // placed copies of this scene instead of the scene itself.  Each transform is the three rows (3x3 part, translation)
    // of an invertible affine matrix.  Needs the flattened layout; an empty vector goes back to the single object.
    void place_copies(const std::vector<float>& object_to_world);
    // nx * nz translated copies on a grid in x and z, 2 % of the scene's extent apart.
    void place_copies_on_grid(uint32_t nx, uint32_t nz);

    // Writes `meta` where the kernels read it (slot 0 of the scene buffer).
    void upload_metadata() const;

    metadata meta;

    Buffer* scene_buffer = nullptr;
    Buffer* indices_buffer = nullptr;
    Buffer* positions_buffer = nullptr;
    Buffer* normals_buffer = nullptr;
    Buffer* uvs_buffer = nullptr;
    Buffer* bvh_buffer = nullptr;
    Buffer* materials_buffer = nullptr;
    Buffer* flat_bvh_buffer = nullptr;  // the flattened layout bound to bvh_buffer (wavefront path), may be null
    Buffer* instances_buffer = nullptr; // the top level over placed copies bound to flat_bvh_buffer, may be null
    Buffer* spheres_buffer = nullptr;   // the shader's `sphere` records (compute.comp:31-36) named by the layout's sphere leaves, may be null
    size_t sphere_count = 0;
    size_t copy_count = 1;

    size_t triangle_count = 0;
    std::string bvh_mode;               // "reference_order" or "sah_partitioned"

private:
    Sampler* sampler = nullptr;
    std::vector<Texture*> textures;
};
static_assert(sizeof(scene::metadata) == 160, "scene metadata is 160 bytes");

#endif  // PTR_SCENE_HPP_
