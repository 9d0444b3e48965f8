#include "scene_source.hpp"

#include <cstdio>
#include <cstring>
#include <cmath>
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

std::vector<scene_source::source_sphere> scene_source::load_spheres(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "r");
    if (!f) throw std::runtime_error("scene_source::load_spheres: cannot open " + path);
    std::vector<source_sphere> out;
    char line[512];
    int number = 0;
    while (std::fgets(line, sizeof line, f)) {
        number++;
        if (char* hash = std::strchr(line, '#')) *hash = 0;
        float v[9];
        const int got = std::sscanf(line, "%f %f %f %f %f %f %f %f %f", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6], &v[7], &v[8]);
        if (got <= 0) continue;  // blank or comment
        if (got != 9 || !(v[3] > 0.f)) {
            std::fclose(f);
            throw std::runtime_error("scene_source::load_spheres: " + path + " line " + std::to_string(number) + ": expected x y z radius r g b metalness roughness, radius > 0");
        }
        source_sphere s{};
        s.center[0] = v[0]; s.center[1] = v[1]; s.center[2] = v[2]; s.radius = v[3];
        s.material.base_color[0] = v[4]; s.material.base_color[1] = v[5]; s.material.base_color[2] = v[6]; s.material.base_color[3] = 1.f;
        s.material.albedo_texture = 0; s.material.mr_texture = 0;
        s.material.metalness = v[7]; s.material.roughness = v[8];
        out.push_back(s);
    }
    std::fclose(f);
    return out;
}

std::vector<scene_source::source_sphere> scene_source::random_spheres(uint32_t seed) {
    // a small generator of its own, so that a seed names the same scene everywhere (the reference called rand())
    uint64_t state = 0x9e3779b97f4a7c15ull ^ ((uint64_t)seed << 17 | seed);
    auto randd = [&]() {
        state = state * 6364136223846793005ull + 1442695040888963407ull;
        return (float)((state >> 40) & 0xffffffu) / 16777216.f;   // [0, 1)
    };
    auto make = [](float x, float y, float z, float radius, float r, float g, float b, float metalness, float roughness) {
        source_sphere s{};
        s.center[0] = x; s.center[1] = y; s.center[2] = z; s.radius = radius;
        s.material.base_color[0] = r; s.material.base_color[1] = g; s.material.base_color[2] = b; s.material.base_color[3] = 1.f;
        s.material.albedo_texture = 0; s.material.mr_texture = 0;
        s.material.metalness = metalness; s.material.roughness = roughness;
        return s;
    };
    std::vector<source_sphere> out;
    // The commented code leaves metalness and roughness at the struct's defaults (1 and 1, include/material.hpp:13-14), and a surface
    // of metalness 1 reflects nothing under today's shader (its specular F0 is taken from the diffuse reflectance, which is 0
    // there).  The book's three material kinds are therefore mapped onto what the shader does render: diffuse = metalness 0,
    // roughness 1; metal = metalness 0 .. 0.5 and roughness 0, as the commented code has it; glass = a smooth half-metal.
    out.push_back(make(0.f, -1000.f, 0.f, 1000.f, 0.5f, 0.5f, 0.5f, 0.f, 1.f));
    const int n = 11;
    for (int a = -n; a < n; a++)
        for (int b = -n; b < n; b++) {
            const float choose = randd();
            const float cx = (float)a + 0.9f * randd(), cy = 0.2f, cz = (float)b + 0.9f * randd();
            const float dx = cx - 4.f, dy = cy - 0.2f, dz = cz;
            if (std::sqrt(dx * dx + dy * dy + dz * dz) <= 0.9f) continue;
            if (choose < 0.8f) {          // diffuse
                const float r = randd() * randd(), g = randd() * randd(), bl = randd() * randd();
                out.push_back(make(cx, cy, cz, 0.2f, r, g, bl, 0.f, 1.f));
            } else if (choose < 0.95f) {  // metal
                const float r = 0.5f + 0.5f * randd(), g = 0.5f + 0.5f * randd(), bl = 0.5f + 0.5f * randd();
                out.push_back(make(cx, cy, cz, 0.2f, r, g, bl, 0.5f * randd(), 0.f));
            } else {                      // glass
                out.push_back(make(cx, cy, cz, 0.2f, 1.f, 1.f, 1.f, 0.5f, 0.f));
            }
        }
    out.push_back(make(0.f, 1.f, 0.f, 1.f, 1.f, 1.f, 1.f, 0.5f, 0.f));      // the book's glass ball
    out.push_back(make(-4.f, 1.f, 0.f, 1.f, 0.4f, 0.2f, 0.1f, 0.f, 1.f));   // its diffuse one
    out.push_back(make(4.f, 1.f, 0.f, 1.f, 0.7f, 0.6f, 0.5f, 0.5f, 0.f));   // its metal one
    return out;
}
