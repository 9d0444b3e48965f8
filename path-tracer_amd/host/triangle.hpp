// Triangle record handed to the BVH builder; same shape as the reference's
// include/triangle.hpp:6-18 (three corners + centroid, 64 bytes).
#ifndef PTR_TRIANGLE_HPP_
#define PTR_TRIANGLE_HPP_

#include "vec3.hpp"

struct triangle {
    triangle() = default;
    triangle(vec3 a, vec3 b, vec3 c) : p1(a), p2(b), p3(c) { center = (p1 + p2 + p3) / 3.f; }

    vec3 p1;
    vec3 p2;
    vec3 p3;
    vec3 center;
};

static_assert(sizeof(triangle) == 64, "triangle is four 16-byte lanes");

#endif  // PTR_TRIANGLE_HPP_
