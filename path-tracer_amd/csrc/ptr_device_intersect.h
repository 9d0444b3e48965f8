// Ray/primitive tests of shaders/include/ray.h as device functions, shared by the megakernel and the
// per-function test kernel.
#pragma once

#include "ptr_device_shading.h"

namespace ptr {

// ray.h:66-78 with the reciprocal direction hoisted (it is a pure function of the ray).
PTR_DEV bool hit_aabb(const float4 lo, const float4 hi, const f3 origin, const f3 inv_dir, float min_t, float max_t) {
    const f3 fv = (mk3(hi.x, hi.y, hi.z) - origin) * inv_dir;
    const f3 nv = (mk3(lo.x, lo.y, lo.z) - origin) * inv_dir;
    const float t1 = min_num(min_num(max_num(fv.x, nv.x), min_num(max_num(fv.y, nv.y), max_num(fv.z, nv.z))), max_t);
    const float t0 = max_num(max_num(max_num(min_num(fv.x, nv.x), max_num(min_num(fv.y, nv.y), min_num(fv.z, nv.z))), min_t), 0.0f);
    return t1 >= t0;
}

// ray.h:137-163.  Accepts exactly what the shader accepts (a NaN passes every rejection test).
PTR_DEV bool hit_triangle_pos(const f3 p0, const f3 p1, const f3 p2, const Ray& r, float max_t, float& t_out, float& u_out, float& v_out) {
    const f3 e1 = p1 - p0;
    const f3 e2 = p2 - p0;
    const f3 ov = r.origin - p0;
    const f3 n = cross(e1, e2);
    const f3 q = cross(ov, r.dir);
    const float d = rcp(dot(r.dir, n));
    const float u = d * dot(-q, e2);
    const float v = d * dot(q, e1);
    const float t = d * dot(-n, ov);
    if (u < 0.0f || u > 1.0f || v < 0.0f || u + v > 1.0f || t < r.min_t || t > max_t) return false;
    t_out = t; u_out = u; v_out = v;
    return true;
}

PTR_DEV bool hit_triangle(const SceneView& sv, uint32_t id, const Ray& r, float max_t, float& t_out, float& u_out, float& v_out) {
    uint32_t i0, i1, i2, mat;
    decode_triangle(sv, id, i0, i1, i2, mat);
    return hit_triangle_pos(load3(sv.positions, i0), load3(sv.positions, i1), load3(sv.positions, i2), r, max_t, t_out, u_out, v_out);
}

// ray.h:105-135: nearer root inside [min_t, max_t], else the farther one; sets only t.
PTR_DEV bool hit_sphere(const Sphere& s, const Ray& r, float& t_out) {
    const f3 oc = r.origin - mk3(s.position.x, s.position.y, s.position.z);
    const float a = dot(r.dir, r.dir);
    const float half_b = dot(r.dir, oc);
    const float c = dot(oc, oc) - s.radius * s.radius;
    const float discriminant = half_b * half_b - a * c;
    if (discriminant < 0.0f) return false;
    const float sd = sqrtf(discriminant);
    float root = (-half_b - sd) / a;
    if (root < r.min_t || root > r.max_t) {
        root = (-half_b + sd) / a;
        if (root < r.min_t || root > r.max_t) return false;
    }
    t_out = root;
    return true;
}

}  // namespace ptr
