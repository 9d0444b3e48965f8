// Axis-aligned box with the reference's surface (include/aabb.hpp:10-61 in the reference).
// An empty box is (+FLT_MAX, -FLT_MAX) so that unions work without a special case.
#ifndef PTR_AABB_HPP_
#define PTR_AABB_HPP_

#include <cstdint>
#include <limits>

#include "triangle.hpp"
#include "vec3.hpp"

class aabb {
public:
    aabb() = default;
    aabb(const vec3& min_point, const vec3& max_point) : minimum(min_point), maximum(max_point) {}
    aabb(const vec3& position, float radius) : minimum(position - vec3(radius)), maximum(position + vec3(radius)) {}
    aabb(const triangle& tri) {
        minimum = tri.p1.min(tri.p2.min(tri.p3));
        maximum = tri.p1.max(tri.p2.max(tri.p3));
    }

    void union_with(const aabb& box) {
        minimum = minimum.min(box.minimum);
        maximum = maximum.max(box.maximum);
    }
    void union_with(const point3& point) {
        minimum = minimum.min(point);
        maximum = maximum.max(point);
    }

    // Strict comparisons: ties fall through to the later axis.
    int32_t maximum_axis() const {
        const vec3 d = diagonal();
        if (d.v[0] > d.v[1] && d.v[0] > d.v[2]) return 0;
        if (d.v[1] > d.v[2]) return 1;
        return 2;
    }

    vec3 diagonal() const { return maximum - minimum; }
    vec3 center() const { return diagonal() / 2.f; }

    float surface_area() const {
        const vec3 d = diagonal();
        return 2.f * (d.v[0] * d.v[1] + d.v[0] * d.v[2] + d.v[1] * d.v[2]);
    }

    vec3 minimum = vec3(std::numeric_limits<float>::max());
    vec3 maximum = vec3(std::numeric_limits<float>::lowest());
};

#endif  // PTR_AABB_HPP_
