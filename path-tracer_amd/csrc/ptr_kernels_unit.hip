// Per-function test kernel: evaluates ONE device function of the render path on an array of argument
// records, so each function can be compared with a CPU evaluation bit for bit (the full-frame tests
// only see them composed).  One lane = one record; records are arrays of 32-bit words.
#pragma clang fp contract(off)

#include <string.h>

#include "ptr_kernels.h"
#include "ptr_device_intersect.h"

namespace ptr {

struct UnitFunction { const char* name; uint32_t in_words, out_words; };

// word layouts (f = float bits, u = uint32):
static const UnitFunction kUnitFunctions[] = {
    {"pcg_hash", 1, 2},             // u seed                                   -> u hash, u seed'
    {"rand", 1, 2},                 // u seed                                   -> f value, u seed'
    {"sin", 1, 1},                  // f x                                      -> f
    {"cos", 1, 1},
    {"pow", 2, 1},                  // f x, f y                                 -> f
    {"srgb_to_linear", 3, 3},       // f rgb                                    -> f rgb
    {"linear_to_srgb", 3, 3},
    {"hit_aabb", 14, 1},            // f min[3], max[3], origin[3], dir[3], min_t, max_t -> u hit
    {"hit_triangle", 17, 8},        // f p0[3], p1[3], p2[3], origin[3], dir[3], min_t, max_t -> u hit, f t, bary[3], normal[3]
    {"hit_sphere", 12, 2},          // f centre[3], radius, origin[3], dir[3], min_t, max_t -> u hit, f t
    {"rotation_to_z_axis", 3, 4},   // f axis[3]                                -> f q[4]
    {"rotate_point", 7, 3},         // f q[4], v[3]                             -> f v'[3]
    {"sample_hemisphere", 2, 3},    // f r0, r1                                 -> f dir[3]
    {"disk_vec", 2, 2},             // f r0, r1                                 -> f xy
    {"sample_ggx_vndf", 6, 3},      // f ve[3], ax, ay, u seed                  -> f h[3]
    {"brdf_probability", 10, 1},    // f colour[3], metalness, view[3], normal[3] -> f p
};
constexpr int kUnitFunctionCount = sizeof(kUnitFunctions) / sizeof(kUnitFunctions[0]);

__global__ __launch_bounds__(64) void ptr_unit_function(int fn, const uint32_t* __restrict__ in, uint32_t in_words,
                                                        uint32_t* __restrict__ out, uint32_t out_words, uint32_t count) {
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= count) return;
    const uint32_t* a = in + (size_t)i * in_words;
    uint32_t* o = out + (size_t)i * out_words;
    auto F = [&](uint32_t k) { return as_f32(a[k]); };
    auto V = [&](uint32_t k) { return mk3(as_f32(a[k]), as_f32(a[k + 1]), as_f32(a[k + 2])); };
    auto put3 = [&](uint32_t k, f3 v) { o[k] = as_u32(v.x); o[k + 1] = as_u32(v.y); o[k + 2] = as_u32(v.z); };
    switch (fn) {
    case 0: { uint32_t s = a[0]; o[0] = pcg_hash(s); o[1] = s; break; }
    case 1: { uint32_t s = a[0]; o[0] = as_u32(rand01(s)); o[1] = s; break; }
    case 2: { float s, c; sincos_poly(F(0), s, c); o[0] = as_u32(s); break; }
    case 3: { float s, c; sincos_poly(F(0), s, c); o[0] = as_u32(c); break; }
    case 4: o[0] = as_u32(pow_pos(F(0), F(1))); break;
    case 5: put3(0, srgb_to_linear(V(0))); break;
    case 6: put3(0, linear_to_srgb(V(0))); break;
    case 7: {
        const f3 d = V(9);
        const f3 inv = mk3(rcp(d.x), rcp(d.y), rcp(d.z));
        o[0] = hit_aabb(make_float4(F(0), F(1), F(2), 0.0f), make_float4(F(3), F(4), F(5), 0.0f), V(6), inv, F(12), F(13)) ? 1u : 0u;
        break;
    }
    case 8: {
        Ray r; r.origin = V(9); r.dir = V(12); r.min_t = F(15); r.max_t = F(16);
        float t = 0.0f, u = 0.0f, v = 0.0f;
        const bool hit = hit_triangle_pos(V(0), V(3), V(6), r, r.max_t, t, u, v);
        o[0] = hit ? 1u : 0u; o[1] = as_u32(t);
        o[2] = as_u32((1.0f - u) - v); o[3] = as_u32(u); o[4] = as_u32(v);
        put3(5, normalize(cross(V(3) - V(0), V(6) - V(0))));
        break;
    }
    case 9: {
        Sphere s; s.position = make_float4(F(0), F(1), F(2), 0.0f); s.radius = F(3);
        Ray r; r.origin = V(4); r.dir = V(7); r.min_t = F(10); r.max_t = F(11);
        float t = 0.0f;
        o[0] = hit_sphere(s, r, t) ? 1u : 0u; o[1] = as_u32(t);
        break;
    }
    case 10: { const f4 q = rotation_to_z_axis(V(0)); o[0] = as_u32(q.x); o[1] = as_u32(q.y); o[2] = as_u32(q.z); o[3] = as_u32(q.w); break; }
    case 11: { f4 q; q.x = F(0); q.y = F(1); q.z = F(2); q.w = F(3); put3(0, rotate_point(q, V(4))); break; }
    case 12: put3(0, sample_hemisphere(F(0), F(1))); break;
    case 13: { const f2 d = disk_vec(F(0), F(1)); o[0] = as_u32(d.x); o[1] = as_u32(d.y); break; }
    case 14: put3(0, sample_ggx_vndf(V(0), F(3), F(4), a[5])); break;
    case 15: o[0] = as_u32(brdf_probability(V(0), F(3), V(4), V(7))); break;
    default: break;
    }
}

int unit_function_lookup(const char* name, uint32_t* in_words, uint32_t* out_words) {
    for (int k = 0; k < kUnitFunctionCount; k++)
        if (strcmp(name, kUnitFunctions[k].name) == 0) {
            *in_words = kUnitFunctions[k].in_words; *out_words = kUnitFunctions[k].out_words;
            return k;
        }
    return -1;
}

hipError_t launch_unit_function(hipStream_t s, int fn, const uint32_t* in, uint32_t in_words, uint32_t* out, uint32_t out_words, uint32_t count) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(ptr_unit_function, dim3((count + 63u) / 64u), dim3(64), 0, s, fn, in, in_words, out, out_words, count);
    return hipGetLastError();
}

}  // namespace ptr
