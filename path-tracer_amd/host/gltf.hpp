// glTF 2.0 loader with the interface of the reference's (include/gltf.hpp:12-54): construct from a path, read the
// node tree from `root_node`.  Reads .gltf (+ external or data-URI buffers and images) and .glb containers.
//
// Where the reference's loader only works for one particular file layout, this one implements what the format
// says, because the arrays it produces are the kernels' input:
//   * accessors honour byteOffset, byteStride, componentType and `normalized` (src/gltf.cpp:104-114 copies the
//     whole buffer view instead); indices may be 8, 16 or 32 bits wide, or absent;
//   * a node's `children` is read as the array it is (src/gltf.cpp:51 converts the array to an integer, which
//     throws for every file that has a hierarchy);
//   * missing NORMAL -> area-weighted vertex normals, missing TEXCOORD_0 -> zeros, missing material -> the
//     format's default material; missing baseColorTexture -> texture number 0 (the kernels' opaque white).
// Like the reference, node transforms are ignored unless `apply_transforms` is set.
#ifndef PTR_GLTF_HPP_
#define PTR_GLTF_HPP_

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "image.hpp"
#include "json.hpp"
#include "mesh.hpp"
#include "scene_source.hpp"

struct gltf_error : std::runtime_error { using std::runtime_error::runtime_error; };

struct node {
    std::vector<node> children;
    Mesh* mesh = nullptr;
    double matrix[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };  // local transform, column-major
};

class gltf {
public:
    explicit gltf(const std::string& filepath);
    gltf(const gltf&) = delete;             // nodes point into `meshes`
    gltf& operator=(const gltf&) = delete;

    node root_node;

    // What src/scene.cpp:84-176 does with the node tree: breadth-first over the nodes, every mesh appended to the
    // flat arrays, one material record per submesh visited, indices made absolute and packed with the material id.
    // wide_indices: -1 = the 4-word index layout only when the packed one cannot hold the scene, 0 = packed (throws
    // when it cannot), 1 = always 4 words.
    scene_source flatten(bool apply_transforms = false, int wide_indices = -1) const;

    // Instancing (an extension): each mesh appears ONCE in the arrays, untransformed, as an object of its own, and every
    // node that shows it becomes a placement with the node's world matrix -- node transforms are honoured without copying
    // geometry.  The scene class renders such a source through a top-level tree (ptr_bvh_bind_instances).
    scene_source flatten_instanced(int wide_indices = -1) const;

    const std::vector<Mesh>& get_meshes() const { return meshes; }
    const std::vector<scene_source::source_material>& get_materials() const { return materials; }
    const std::vector<raw_image>& get_images() const { return images; }

private:
    void load_node(uint32_t index, node& parent, int depth);
    void load_meshes();
    Mesh::submesh load_primitive(const json& primitive);
    std::vector<double> load_attribute(uint32_t accessor_index, uint32_t expected_components, uint32_t* count) const;
    void load_buffers(const std::string& parent_path);
    void load_textures(const std::string& parent_path);
    void load_materials();

    json gltf_json;
    std::vector<Mesh> meshes;
    std::vector<scene_source::source_material> materials;  // texture numbers: 1 + index into `textures`, 0 = none
    std::vector<uint32_t> textures;                        // glTF texture -> index into `images`
    std::vector<raw_image> images;
    std::vector<std::vector<uint8_t>> buffers;
    std::vector<uint8_t> glb_binary;                       // BIN chunk of a .glb container
    uint32_t default_material = 0xffffffffu;               // created on first use
};

#endif  // PTR_GLTF_HPP_
