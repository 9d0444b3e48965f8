// fp32 device arithmetic of the render path.
//
// The kernels must produce the same bits as a scalar CPU evaluation of the shader, so nothing here
// may depend on hardware approximations (v_rcp / v_rsq / v_sin / v_exp / v_log) or on FMA contraction:
//   * + - * / sqrt are IEEE round-to-nearest (build with -ffp-contract=off and correctly rounded
//     divide/sqrt, which is hipcc's default for fp32), denormals kept;
//   * min/max are v_min_f32 / v_max_f32 under IEEE mode = minimumNumber / maximumNumber;
//   * normalize(v) = v * (1 / sqrt(dot(v, v))); dot products sum left to right;
//   * sin/cos/exp2/log2 are polynomial evaluations written out below.
#pragma once
#pragma clang fp contract(off)

#include <hip/hip_runtime.h>
#include <stdint.h>

#define PTR_DEV __device__ __forceinline__

namespace ptr {

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

PTR_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
PTR_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
PTR_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
PTR_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
PTR_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
PTR_DEV f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
PTR_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
PTR_DEV float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PTR_DEV f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
PTR_DEV float rcp(float x) { return 1.0f / x; }
PTR_DEV f3 normalize(f3 a) { return a * rcp(sqrtf(dot(a, a))); }

PTR_DEV float min_num(float a, float b) { return __builtin_fminf(a, b); }
PTR_DEV float max_num(float a, float b) { return __builtin_fmaxf(a, b); }
PTR_DEV float clampf(float x, float lo, float hi) { return min_num(max_num(x, lo), hi); }
PTR_DEV float mixf(float a, float b, float w) { return a * (1.0f - w) + b * w; }

PTR_DEV uint32_t as_u32(float f) { return __float_as_uint(f); }
PTR_DEV float as_f32(uint32_t u) { return __uint_as_float(u); }

// sin and cos of x >= 0 together.  Quadrant q = round(x * 2/pi); r = x - q*pi/2 with pi/2 split into
// three fp32 terms whose leading products are exact; odd/even minimax polynomials on |r| <= pi/4.
PTR_DEV void sincos_poly(float x, float& s, float& c) {
    const int q = (int)(x * 0.63661977236758134f + 0.5f);
    const float qf = (float)q;
    const float r = ((x - qf * 1.5703125f) - qf * 4.837512969970703125e-4f) - qf * 7.54978995489188e-8f;
    const float z = r * r;
    const float sr = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    const float cr = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    const bool swap = (q & 1) != 0;
    const float sv = swap ? cr : sr;
    const float cv = swap ? sr : cr;
    s = (q & 2) ? -sv : sv;
    c = (((q + 1) & 2) != 0) ? -cv : cv;
}

// log2(x), x > 0: x = 2^e * m with m in [sqrt(1/2), sqrt(2)), log(m) from the s = f/(2+f) series.
PTR_DEV float log2_poly(float x) {
    int e_adj = 0;
    if (x < 1.17549435e-38f) { x = x * 8388608.0f; e_adj = -23; }
    const uint32_t ix = as_u32(x);
    const int32_t e = (int32_t)(ix - 0x3f3504f3u) >> 23;
    const float m = as_f32(ix - ((uint32_t)e << 23));
    const float f = m - 1.0f;
    const float s = f / (2.0f + f);
    const float z = s * s;
    const float w = z * z;
    const float t1 = w * (0.40000972152f + w * 0.24279078841f);
    const float t2 = z * (0.66666662693f + w * 0.28498786688f);
    const float R = t2 + t1;
    const float hfsq = 0.5f * f * f;
    const float ln_m = f - (hfsq - s * (hfsq + R));
    return ln_m * 1.44269504089f + (float)(e + e_adj);
}

// 2^y: y = k + r, |r| <= 1/2, degree-6 polynomial, 2^k applied through the exponent field.
PTR_DEV float exp2_poly(float y) {
    if (!(y > -150.0f)) return 0.0f;
    if (y > 127.0f) y = 127.0f;
    const float kf = floorf(y + 0.5f);
    const float r = y - kf;
    const int k = (int)kf;
    float p = 1.535336188319500e-4f;
    p = p * r + 1.339887440266574e-3f;
    p = p * r + 9.618437357674640e-3f;
    p = p * r + 5.550332471162809e-2f;
    p = p * r + 2.402264791363012e-1f;
    p = p * r + 6.931472028550421e-1f;
    p = p * r + 1.0f;
    if (k >= -126) return p * as_f32((uint32_t)(k + 127) << 23);
    return (p * as_f32((uint32_t)(k + 64 + 127) << 23)) * 5.42101086e-20f;
}

// pow for the sRGB transfer curves: base in [0, ~1.06], positive exponent.
PTR_DEV float pow_pos(float x, float y) {
    if (!(x > 0.0f)) return 0.0f;
    return exp2_poly(y * log2_poly(x));
}
PTR_DEV float pow5(float x) { const float x2 = x * x; return (x2 * x2) * x; }

}  // namespace ptr
