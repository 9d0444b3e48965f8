#include "mesh.hpp"

#include <stdexcept>

// src/mesh.cpp:5-37.  Every attribute array always grows by the submesh's vertex count (a missing attribute is
// zero-filled by the loader before it gets here), so a vertex has the same number in all three arrays.
void Mesh::add_submesh(const submesh& submesh, uint32_t material) {
    if (submesh.index_count % 3 || submesh.indices.size() != submesh.index_count) throw std::runtime_error("Mesh::add_submesh: index count must be a multiple of 3");
    submeshes.push_back(submesh_info{ vertex_count, index_count, submesh.index_count / 3, material });
    indices.insert(indices.end(), submesh.indices.begin(), submesh.indices.end());
    for (uint32_t a = ATTRIBUTE_TYPE::POSITION; a < ATTRIBUTE_TYPE::MAX_ATTRIBUTE; a++) {
        const auto& src = submesh.attributes[a];
        const size_t want = (size_t)submesh.vertex_count * attribute_components(static_cast<ATTRIBUTE_TYPE>(a));
        if (src.size() != want) throw std::runtime_error("Mesh::add_submesh: attribute size does not match the vertex count");
        attributes[a].insert(attributes[a].end(), src.begin(), src.end());
    }
    vertex_count += submesh.vertex_count;
    index_count += submesh.index_count;
}
