// The flat scene arrays the kernels read, before upload: what src/scene.cpp:84-176 produces from a glTF node tree.
// Filled by the glTF loader (gltf::flatten) or read from a .ptrscene file.  No GPU dependency.
#ifndef PTR_SCENE_SOURCE_HPP_
#define PTR_SCENE_SOURCE_HPP_

#include <cstdint>
#include <string>
#include <vector>

struct scene_source {
    // 3 words per triangle: 24-bit vertex index | one byte of the 24-bit material id (src/scene.cpp:140-142), or,
    // for scenes past 2^24 vertices or materials (an extension), 4 plain words per triangle: v0, v1, v2, material
    uint32_t index_words = 3;
    std::vector<uint32_t> indices;
    size_t triangle_count() const { return indices.size() / index_words; }
    std::vector<float> positions;      // 3 per vertex
    std::vector<float> normals;        // 3 per vertex
    std::vector<float> uvs;            // 2 per vertex
    struct source_material { float base_color[4]; uint32_t albedo_texture, mr_texture; float metalness, roughness; };
    std::vector<source_material> materials;  // texture numbers are 1-based indices into `textures`, 0 = none
    struct source_texture { uint32_t width, height; std::vector<uint8_t> rgba; };
    std::vector<source_texture> textures;

    // Instancing (optional, an extension): triangles [object_first[k], object_first[k + 1]) form object k, and the scene
    // is the placements -- copies of objects through affine object-to-world matrices (rows of 3x3 part, translation) --
    // instead of the triangle soup itself.  Empty = no instancing.
    std::vector<uint32_t> object_first;
    struct placement { uint32_t object; float object_to_world[12]; };
    std::vector<placement> placements;

    // Spheres beside (or instead of) the triangles: the primitive of the reference's first scene (include/sphere.hpp:7-19), rendered
    // through sphere leaves of the flattened layout.  Materials are given inline, as the shader's `sphere` struct holds them.
    struct source_sphere { float center[3]; float radius; source_material material; };
    std::vector<source_sphere> spheres;

    static scene_source load(const std::string& path);  // .ptrscene file, see scene_source.cpp
    // One sphere per line: x y z radius r g b metalness roughness ('#' starts a comment).
    static std::vector<source_sphere> load_spheres(const std::string& path);
    // The scene the reference's README picture shows, after its commented-out scene::random_scene (src/scene.cpp:11-58): a ground
    // sphere of radius 1000, a 22 x 22 field of small spheres with random diffuse / metal / mirror materials, three large ones.
    // Deterministic in `seed` (the reference drew from rand()).
    static std::vector<source_sphere> random_spheres(uint32_t seed);
};

#endif  // PTR_SCENE_SOURCE_HPP_
