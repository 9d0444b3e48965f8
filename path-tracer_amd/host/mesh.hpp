// Mesh: the vertex attributes and indices of one glTF mesh, submeshes (= glTF primitives) appended one after
// the other (include/mesh.hpp:18-99 in the reference).  Differences on purpose: indices are kept as uint32 (the
// reference stores uint16 and therefore cannot hold primitives indexed with 32-bit integers), a submesh refers
// to its material by number instead of by reference, and there are no GPU-side members.
#pragma once

#include <array>
#include <cstddef>
#include <cstdint>
#include <vector>

enum ATTRIBUTE_TYPE {
    POSITION,
    NORMAL,
    UV_0,
    MAX_ATTRIBUTE,
};

class Mesh {
public:
    struct submesh_info {
        uint32_t vertex_offset;   // first vertex of the submesh inside the mesh
        uint32_t index_offset;    // first index of the submesh inside the mesh
        size_t element_count;     // triangles
        uint32_t material;        // number of the material in the file's material list
    };

    struct mesh_instance {
        const Mesh& mesh;
    };

    // One decoded primitive: tightly packed float attributes (3, 3, 2 components) and submesh-relative indices.
    struct submesh {
        std::vector<uint32_t> indices;
        std::array<std::vector<float>, ATTRIBUTE_TYPE::MAX_ATTRIBUTE> attributes;
        uint32_t vertex_count = 0;
        uint32_t index_count = 0;
    };

    Mesh() = default;

    [[nodiscard]] mesh_instance instantiate() const { return { *this }; }

    void add_submesh(const submesh& submesh, uint32_t material);

    static inline uint32_t attribute_components(enum ATTRIBUTE_TYPE attribute_type) {
        switch (attribute_type) {
            case ATTRIBUTE_TYPE::POSITION:
            case ATTRIBUTE_TYPE::NORMAL:
                return 3;
            case ATTRIBUTE_TYPE::UV_0:
                return 2;
            default:
                return 0;
        }
    }

    static inline size_t attribute_size(enum ATTRIBUTE_TYPE attribute_type) { return sizeof(float) * attribute_components(attribute_type); }

    [[nodiscard]] inline uint32_t triangle_count() const { return index_count / 3; }
    [[nodiscard]] inline uint32_t vertices_count() const { return vertex_count; }
    [[nodiscard]] inline const std::vector<submesh_info>& get_submeshes() const { return submeshes; }
    [[nodiscard]] inline const std::vector<float>& get_attribute(enum ATTRIBUTE_TYPE attribute_type) const { return attributes[(uint32_t)attribute_type]; }
    [[nodiscard]] inline const std::vector<uint32_t>& get_indices() const { return indices; }

private:
    uint32_t vertex_count = 0;
    uint32_t index_count = 0;

    std::array<std::vector<float>, ATTRIBUTE_TYPE::MAX_ATTRIBUTE> attributes;
    std::vector<uint32_t> indices;
    std::vector<submesh_info> submeshes;
};
