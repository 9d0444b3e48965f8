#include "scene_source.hpp"

#include <cstdio>
#include <cstring>
#include <stdexcept>

// .ptrscene: little-endian, "PTRS", u32 version = 1 (3 index words per triangle) or 2 (4 words: v0, v1, v2, material), u32 index_words, u32 vertices, u32 materials, u32 textures,
// then indices, positions, normals, uvs, materials (48 bytes: base_color[4], albedo texture, unused, mr texture,
// unused, metalness, roughness, pad[2]), then per texture u32 width, u32 height, RGBA8 texels.
scene_source scene_source::load(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("scene_source::load: cannot open " + path);
    auto read = [&](void* dst, size_t n) { if (n && std::fread(dst, 1, n, f) != n) { std::fclose(f); throw std::runtime_error("scene_source::load: truncated " + path); } };
    uint32_t head[6];
    read(head, sizeof head);
    if (std::memcmp(&head[0], "PTRS", 4) != 0 || (head[1] != 1 && head[1] != 2)) { std::fclose(f); throw std::runtime_error("scene_source::load: not a .ptrscene v1/v2 file"); }
    scene_source s;
    s.index_words = head[1] == 2 ? 4u : 3u;
    s.indices.resize(head[2]);
    s.positions.resize((size_t)head[3] * 3);
    s.normals.resize((size_t)head[3] * 3);
    s.uvs.resize((size_t)head[3] * 2);
    read(s.indices.data(), s.indices.size() * 4);
    read(s.positions.data(), s.positions.size() * 4);
    read(s.normals.data(), s.normals.size() * 4);
    read(s.uvs.data(), s.uvs.size() * 4);
    for (uint32_t i = 0; i < head[4]; i++) {
        struct { float base_color[4]; uint32_t albedo, albedo_sampler, mr, mr_sampler; float metalness, roughness, pad[2]; } m;
        read(&m, sizeof m);
        source_material sm;
        std::memcpy(sm.base_color, m.base_color, sizeof sm.base_color);
        sm.albedo_texture = m.albedo; sm.mr_texture = m.mr; sm.metalness = m.metalness; sm.roughness = m.roughness;
        s.materials.push_back(sm);
    }
    for (uint32_t i = 0; i < head[5]; i++) {
        uint32_t wh[2];
        read(wh, sizeof wh);
        source_texture t;
        t.width = wh[0]; t.height = wh[1];
        t.rgba.resize((size_t)wh[0] * wh[1] * 4);
        read(t.rgba.data(), t.rgba.size());
        s.textures.push_back(std::move(t));
    }
    std::fclose(f);
    return s;
}
