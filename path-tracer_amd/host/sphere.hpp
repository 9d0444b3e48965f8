// The sphere primitive the reference still declares (include/sphere.hpp:7-19) although its scene of spheres is commented out
// (src/scene.cpp:11-58), and the record the shader declares for it (shaders/compute.comp:31-36).
#ifndef PTR_SPHERE_HPP_
#define PTR_SPHERE_HPP_

#include "aabb.hpp"
#include "material.hpp"

struct sphere {
    sphere() = default;
    sphere(point3 center, material material, float radius) : center(center), mat(material), radius(radius) {}

    point3 center;
    material mat;
    float radius = 0.f;
    float padding[3] = { 0.f, 0.f, 0.f };
};

// What the kernels read: vec4 position, the 48-byte material, float radius, padded to a multiple of 16 bytes.
struct gpu_sphere {
    float position[4] = { 0.f, 0.f, 0.f, 0.f };
    gpu_material mat;
    float radius = 0.f;
    float padding[3] = { 0.f, 0.f, 0.f };
};
static_assert(sizeof(gpu_sphere) == 80, "gpu_sphere is 80 bytes");

#endif  // PTR_SPHERE_HPP_
