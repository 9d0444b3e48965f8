// Host and device material records (include/material.hpp:9-25 in the reference).  gpu_material is the
// 48-byte record the kernels read.
#pragma once

#include <cstdint>

#include "vec3.hpp"

class Texture;

struct material {
    color base_color;
    Texture* base_color_texture = nullptr;
    Texture* metallic_roughness_texture = nullptr;
    float metalness = 1.f;
    float roughness = 1.f;
};

struct gpu_material {
    color base_color;
    uint32_t albedo_texture_id = 0;
    uint32_t albedo_texture_sampler_id = 0;
    uint32_t metallic_roughness_texture_id = 0;
    uint32_t metallic_roughness_texture_sampler_id = 0;
    float metalness = 1.f;
    float roughness = 1.f;
    float padding[2] = { 0.f, 0.f };
};
static_assert(sizeof(gpu_material) == 48, "gpu_material is 48 bytes");
