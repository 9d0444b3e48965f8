// 16-byte vec3 with the reference's public surface (include/vec3.hpp:6-73 in the reference):
// same member names, same `v[i]` lane access, same 16-byte size so that `camera` stays a
// 128-byte block that can be memcpy'd into the scene metadata.
//
// Plain scalar C++ instead of SSE intrinsics.  Every operation reproduces the rounding order
// of the reference's SSE formulation, which is what makes bvh/camera output bit-identical:
//   * min/max follow MINPS/MAXPS semantics (`a < b ? a : b`, second operand on NaN/equal),
//   * dot is (x*x' + y*y') + (z*z' + w*w')  -- the DPPS pairwise order, lane 3 included,
//   * division is a true division per lane, never a multiply by reciprocal.
// Build with -ffp-contract=off.
#ifndef PTR_VEC3_HPP_
#define PTR_VEC3_HPP_

#include <cmath>
#include <cstdio>
#include <limits>

class alignas(16) vec3 {
public:
    vec3() : v{ 0.f, 0.f, 0.f, 0.f } {}
    vec3(float x) : v{ x, x, x, x } {}  // all four lanes, like _mm_set1_ps
    vec3(float x, float y, float z) : v{ x, y, z, 0.f } {}

    vec3 operator-() const { return from(-v[0], -v[1], -v[2], -v[3]); }

    vec3& operator+=(const vec3& o) {
        for (int i = 0; i < 4; i++) v[i] += o.v[i];
        return *this;
    }
    vec3& operator*=(float s) {
        for (int i = 0; i < 4; i++) v[i] *= s;
        return *this;
    }
    vec3& operator/=(float s) {
        for (int i = 0; i < 4; i++) v[i] /= s;
        return *this;
    }

    float operator[](int axis) const {
        if (axis > 2) return std::numeric_limits<float>::infinity();
        return v[axis];
    }

    float dot(const vec3& o) const { return (v[0] * o.v[0] + v[1] * o.v[1]) + (v[2] * o.v[2] + v[3] * o.v[3]); }
    float length_sq() const { return dot(*this); }
    float length() const { return std::sqrt(length_sq()); }

    vec3 cross(const vec3& o) const {
        return from(v[1] * o.v[2] - v[2] * o.v[1], v[2] * o.v[0] - v[0] * o.v[2], v[0] * o.v[1] - v[1] * o.v[0],
                    v[3] * o.v[3] - v[3] * o.v[3]);
    }

    vec3 min(const vec3& o) const {
        return from(v[0] < o.v[0] ? v[0] : o.v[0], v[1] < o.v[1] ? v[1] : o.v[1], v[2] < o.v[2] ? v[2] : o.v[2],
                    v[3] < o.v[3] ? v[3] : o.v[3]);
    }
    vec3 max(const vec3& o) const {
        return from(v[0] > o.v[0] ? v[0] : o.v[0], v[1] > o.v[1] ? v[1] : o.v[1], v[2] > o.v[2] ? v[2] : o.v[2],
                    v[3] > o.v[3] ? v[3] : o.v[3]);
    }

    vec3& normalize() {
        const float len = std::sqrt(dot(*this));
        for (int i = 0; i < 4; i++) v[i] /= len;
        return *this;
    }

    bool near_zero() const {
        const double eps = 0.000001;
        return (std::fabs(v[0]) < eps) && (std::fabs(v[1]) < eps) && (std::fabs(v[2]) < eps);
    }

    void print() const { std::printf("%f %f %f\n", v[0], v[1], v[2]); }

    static vec3 from(float x, float y, float z, float w) {
        vec3 r;
        r.v[0] = x; r.v[1] = y; r.v[2] = z; r.v[3] = w;
        return r;
    }

    float v[4];
};

inline vec3 operator+(const vec3& a, const vec3& b) { return vec3::from(a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2], a.v[3] + b.v[3]); }
inline vec3 operator-(const vec3& a, const vec3& b) { return vec3::from(a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2], a.v[3] - b.v[3]); }
inline vec3 operator*(const vec3& a, const vec3& b) { return vec3::from(a.v[0] * b.v[0], a.v[1] * b.v[1], a.v[2] * b.v[2], a.v[3] * b.v[3]); }
inline vec3 operator*(const vec3& a, float s) { return vec3::from(a.v[0] * s, a.v[1] * s, a.v[2] * s, a.v[3] * s); }
inline vec3 operator*(float s, const vec3& a) { return a * s; }
inline vec3 operator/(const vec3& a, float s) { return vec3::from(a.v[0] / s, a.v[1] / s, a.v[2] / s, a.v[3] / s); }

inline vec3 lerp(const vec3& u, const vec3& w, float t) { return (1.f - t) * u + t * w; }
inline vec3 reflect(const vec3& d, const vec3& n) { return d - 2 * d.dot(n) * n; }

using point3 = vec3;
using color = vec3;

static_assert(sizeof(vec3) == 16, "vec3 must stay 16 bytes");

#endif  // PTR_VEC3_HPP_
