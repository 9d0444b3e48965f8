/* TEST INFRASTRUCTURE ONLY -- see oracle.h for the contract and the parity status.
 * Build: gcc -O2 -std=c11 -ffp-contract=off (oracle/Makefile).  No product code is included here. */
#define _GNU_SOURCE
#include "oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * fp32 building blocks
 * ---------------------------------------------------------------------------------------------- */
typedef struct { float x, y, z; } v3;
typedef struct { float x, y; } v2;
typedef struct { float x, y, z, w; } v4;

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline int is_nan(float a) { return a != a; }
static inline int is_neg(float a) { return (int)(f2u(a) >> 31); }

/* IEEE minimumNumber / maximumNumber: NaN operand ignored, -0 < +0 */
static inline float fmin_(float a, float b) {
    if (is_nan(a)) return b;
    if (is_nan(b)) return a;
    if (a == b) return is_neg(a) ? a : b;
    return a < b ? a : b;
}
static inline float fmax_(float a, float b) {
    if (is_nan(a)) return b;
    if (is_nan(b)) return a;
    if (a == b) return is_neg(a) ? b : a;
    return a > b ? a : b;
}
static inline float clamp_(float x, float lo, float hi) { return fmin_(fmax_(x, lo), hi); }
static inline float mix_(float a, float b, float w) { return a * (1.0f - w) + b * w; }

static inline v3 V3(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul3(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scale3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 div3s(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline v3 neg3(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 cross3(v3 a, v3 b) {
    return V3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
static inline v3 normalize3(v3 a) { return scale3(a, 1.0f / sqrtf(dot3(a, a))); }

/* --- sin / cos for x >= 0 (the path only evaluates them at 2*pi*rand): quadrant by Cody-Waite
 *     reduction against pi/2 split in three fp32 pieces, then degree-7 / degree-8 polynomials. --- */
static void sincos_(float x, float* s_out, float* c_out) {
    const float two_over_pi = 0.63661977236758134f;
    const float p1 = 1.5703125f;                 /* pi/2 split: products q*p1, q*p2 are exact for small q */
    const float p2 = 4.837512969970703125e-4f;
    const float p3 = 7.54978995489188e-8f;
    int q = (int)(x * two_over_pi + 0.5f);
    float qf = (float)q;
    float r = ((x - qf * p1) - qf * p2) - qf * p3;
    float z = r * r;
    float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    switch (q & 3) {
    case 0: *s_out = sp; *c_out = cp; break;
    case 1: *s_out = cp; *c_out = -sp; break;
    case 2: *s_out = -sp; *c_out = -cp; break;
    default: *s_out = -cp; *c_out = sp; break;
    }
}

/* --- log2 for x > 0: x = 2^e * m, m in [sqrt(1/2), sqrt(2)); log(m) by the classic s = f/(2+f) series --- */
static float log2_(float x) {
    int e_adj = 0;
    if (x < 1.17549435e-38f) { x = x * 8388608.0f; e_adj = -23; } /* denormal input */
    uint32_t ix = f2u(x);
    int32_t e = (int32_t)(ix - 0x3f3504f3u) >> 23; /* arithmetic shift: floor(log2(x / sqrt(1/2))) */
    float m = u2f(ix - ((uint32_t)e << 23));
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float w = z * z;
    float t1 = w * (0.40000972152f + w * 0.24279078841f);
    float t2 = z * (0.66666662693f + w * 0.28498786688f);
    float R = t2 + t1;
    float hfsq = 0.5f * f * f;
    float ln_m = f - (hfsq - s * (hfsq + R));
    return ln_m * 1.44269504089f + (float)(e + e_adj);
}

/* --- exp2: y = k + r, |r| <= 1/2, degree-6 polynomial, scale by 2^k through the exponent field --- */
static float exp2_(float y) {
    if (!(y > -150.0f)) return 0.0f; /* also NaN -> 0 */
    if (y > 127.0f) y = 127.0f;
    float kf = floorf(y + 0.5f);
    float r = y - kf;
    int k = (int)kf;
    float p = 1.535336188319500e-4f;
    p = p * r + 1.339887440266574e-3f;
    p = p * r + 9.618437357674640e-3f;
    p = p * r + 5.550332471162809e-2f;
    p = p * r + 2.402264791363012e-1f;
    p = p * r + 6.931472028550421e-1f;
    p = p * r + 1.0f;
    if (k >= -126) return p * u2f((uint32_t)(k + 127) << 23);
    return (p * u2f((uint32_t)(k + 64 + 127) << 23)) * 5.42101086e-20f; /* 2^-64: lands in the denormals */
}

static float pow_(float x, float y) {
    if (!(x > 0.0f)) return 0.0f; /* the path only raises values in [0, ~1.06] to positive powers */
    return exp2_(y * log2_(x));
}
static inline float pow5_(float x) { float x2 = x * x; return (x2 * x2) * x; }

/* ------------------------------------------------------------------------------------------------
 * rand.h:1-12
 * ---------------------------------------------------------------------------------------------- */
uint32_t orc_pcg_hash(uint32_t* seed) {
    uint32_t state = *seed;
    *seed = state * (747796405u + 2891336453u); /* `seed *= a + b`: the sum is the multiplier, no increment */
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
float orc_rand(uint32_t* seed) { return (float)orc_pcg_hash(seed) / 4294967296.0f; }

/* ------------------------------------------------------------------------------------------------
 * color_utils.h:1-26
 * ---------------------------------------------------------------------------------------------- */
static inline float less_than_(float v, float value) { return v < value ? 1.0f : 0.0f; }
static float linear_to_srgb1(float c) {
    c = clamp_(c, 0.0f, 1.0f);
    return mix_(pow_(c, 1.0f / 2.4f) * 1.055f - 0.055f, c * 12.92f, less_than_(c, 0.0031308f));
}
static float srgb_to_linear1(float c) {
    c = clamp_(c, 0.0f, 1.0f);
    return mix_(pow_((c + 0.055f) / 1.055f, 2.4f), c / 12.92f, less_than_(c, 0.04045f));
}
static v3 linear_to_srgb(v3 c) { return V3(linear_to_srgb1(c.x), linear_to_srgb1(c.y), linear_to_srgb1(c.z)); }
static v3 srgb_to_linear(v3 c) { return V3(srgb_to_linear1(c.x), srgb_to_linear1(c.y), srgb_to_linear1(c.z)); }
static inline float luminance(v3 c) { return dot3(c, V3(0.2126f, 0.7152f, 0.0722f)); }

/* ------------------------------------------------------------------------------------------------
 * math.h:1-53
 * ---------------------------------------------------------------------------------------------- */
#define PI_F 3.141592653589793f
#define TWO_PI_F (PI_F * 2.0f)

static v3 sample_hemisphere(float r0, float r1) { /* math.h:12-17 */
    float radius = sqrtf(r0);
    float theta = 2.0f * PI_F * r1;
    float s, c;
    sincos_(theta, &s, &c);
    return V3(c * radius, s * radius, sqrtf(1.0f - r0));
}
static v2 disk_vec(float r0, float r1) { /* math.h:19-25 */
    float z = r0 * 2.0f - 1.0f;
    float a = 2.0f * PI_F * r1;
    float r = sqrtf(1.0f - z * z);
    float s, c;
    sincos_(a, &s, &c);
    v2 o = { c * r, s * r };
    return o;
}
static v4 rotation_to_z_axis(v3 axis) { /* math.h:29-34 */
    v4 q;
    if (axis.z < -0.99999f) { q.x = 1.0f; q.y = 0.0f; q.z = 0.0f; q.w = 0.0f; return q; }
    float x = axis.y, y = -axis.x, z = 0.0f, w = 1.0f + axis.z;
    float inv = 1.0f / sqrtf(x * x + y * y + z * z + w * w);
    q.x = x * inv; q.y = y * inv; q.z = z * inv; q.w = w * inv;
    return q;
}
static inline v4 invert_rotation(v4 q) { v4 r = { -q.x, -q.y, -q.z, q.w }; return r; } /* math.h:36-38 */
static v3 rotate_point(v4 q, v3 v) { /* math.h:42-45 */
    v3 qa = V3(q.x, q.y, q.z);
    v3 a = scale3(qa, 2.0f * dot3(qa, v));
    v3 b = scale3(v, q.w * q.w - dot3(qa, qa));
    v3 c = scale3(cross3(qa, v), 2.0f * q.w);
    return add3(add3(a, b), c);
}
static inline v3 interp3(const v3 val[3], v3 w) { /* math.h:47-49 */
    return add3(add3(scale3(val[0], w.x), scale3(val[1], w.y)), scale3(val[2], w.z));
}
static inline v2 interp2(const v2 val[3], v3 w) { /* math.h:51-53 */
    v2 r = { val[0].x * w.x + val[1].x * w.y + val[2].x * w.z, val[0].y * w.x + val[1].y * w.y + val[2].y * w.z };
    return r;
}

/* ------------------------------------------------------------------------------------------------
 * brdf.h:1-99
 * ---------------------------------------------------------------------------------------------- */
#define MIN_DIELECTRICS_F0 0.04f

static v3 sample_ggx_vndf(v3 Ve, float ax, float ay, uint32_t seed /* by value: brdf.h:13 */) {
    v3 Vh = normalize3(V3(ax * Ve.x, ay * Ve.y, Ve.z));
    float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    v3 T1 = lensq > 0.0f ? scale3(V3(-Vh.y, Vh.x, 0.0f), 1.0f / sqrtf(lensq)) : V3(1.0f, 0.0f, 0.0f);
    v3 T2 = cross3(Vh, T1);
    float r = sqrtf(orc_rand(&seed));
    float phi = TWO_PI_F * orc_rand(&seed);
    float sphi, cphi;
    sincos_(phi, &sphi, &cphi);
    float t1 = r * cphi;
    float t2 = r * sphi;
    float s = 0.5f * (1.0f + Vh.z);
    t2 = mix_(sqrtf(1.0f - t1 * t1), t2, s);
    v3 Nh = add3(add3(scale3(T1, t1), scale3(T2, t2)), scale3(Vh, sqrtf(fmax_(0.0f, 1.0f - t1 * t1 - t2 * t2))));
    return normalize3(V3(ax * Nh.x, ay * Nh.y, fmax_(0.0f, Nh.z)));
}
static float smith_g1_ggx(float alpha, float NdotS, float alpha_squared, float NdotS_squared) { /* brdf.h:38-40 */
    (void)alpha; (void)NdotS;
    return 2.0f / (sqrtf(((alpha_squared * (1.0f - NdotS_squared)) + NdotS_squared) / NdotS_squared) + 1.0f);
}
static inline v3 base_color_to_specular_f0(v3 c, float metalness) { /* brdf.h:51-53 */
    return V3(mix_(MIN_DIELECTRICS_F0, c.x, metalness), mix_(MIN_DIELECTRICS_F0, c.y, metalness),
              mix_(MIN_DIELECTRICS_F0, c.z, metalness));
}
static inline v3 base_color_to_diffuse_reflectance(v3 c, float metalness) { /* brdf.h:55-57 */
    return scale3(c, 1.0f - metalness);
}
static inline v3 eval_fresnel_schlick(v3 f0, float f90, float NdotS) { /* brdf.h:61-63 */
    float p = pow5_(1.0f - NdotS);
    return V3(f0.x + (f90 - f0.x) * p, f0.y + (f90 - f0.y) * p, f0.z + (f90 - f0.z) * p);
}
static inline float shadowed_f90(v3 F0) { /* brdf.h:75-81 */
    const float t = 1.0f / MIN_DIELECTRICS_F0;
    return fmin_(1.0f, t * luminance(F0));
}
static float get_brdf_probability(v3 color, float metalness, v3 view, v3 shading_normal) { /* brdf.h:83-99 */
    float l = luminance(base_color_to_specular_f0(color, metalness));
    v3 specular_F0 = V3(l, l, l);
    float diffuse_reflectance = luminance(base_color_to_diffuse_reflectance(color, metalness));
    float fresnel = clamp_(
        luminance(eval_fresnel_schlick(specular_F0, shadowed_f90(specular_F0), fmax_(0.0f, dot3(view, shading_normal)))),
        0.0f, 1.0f);
    float specular = fresnel;
    float diffuse = diffuse_reflectance * (1.0f - fresnel);
    float p = specular / fmax_(0.0001f, specular + diffuse);
    return clamp_(p, 0.1f, 0.9f);
}

/* ------------------------------------------------------------------------------------------------
 * scene data access (compute.comp:21-120, ray.h:20-64)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    float base_color[4];
    uint32_t albedo_tex, albedo_sampler, mr_tex, mr_sampler;
    float metalness, roughness;
    float pad[2];
} material_t; /* 48 bytes */

typedef struct {
    float min[3]; int32_t next_id;
    float max[3]; int32_t primitive_id;
} node_t; /* 32 bytes */

typedef struct {
    /* camera block as the SHADER declares it (compute.comp:45-58): `up` at 32, `right` at 48 */
    float position[4], forward[4], up[4], right[4], horizontal[4], vertical[4], first_pixel[4];
    float lens_radius, cam_pad[3];
    uint32_t max_bounce, min_bounce, width, height, sample_index, enable_dof, debug_bvh;
    int32_t downscale_factor;
} meta_t; /* 160 bytes */

typedef struct { v3 origin, direction; float min_t, max_t; } ray_t;
typedef struct { v3 point, barycentrics, geometry_normal; float t; uint32_t primitive_id; int32_t instance; } hit_t;

static inline void decode_triangle_indices(const orc_frame* f, uint32_t tri, uint32_t out[4]) { /* ray.h:20-30 */
    if (f->index_words == 4) { /* wide extension: plain vertex ids and material id */
        const uint32_t* w = f->indices + (size_t)tri * 4;
        out[0] = w[0]; out[1] = w[1]; out[2] = w[2]; out[3] = w[3];
        return;
    }
    uint32_t a = f->indices[(size_t)tri * 3], b = f->indices[(size_t)tri * 3 + 1], c = f->indices[(size_t)tri * 3 + 2];
    out[0] = a & 0x00ffffffu;
    out[1] = b & 0x00ffffffu;
    out[2] = c & 0x00ffffffu;
    out[3] = ((a & 0xff000000u) >> 8) | ((b & 0xff000000u) >> 16) | ((c & 0xff000000u) >> 24);
}
static inline v3 load3(const float* p, uint32_t i) { return V3(p[i * 3], p[i * 3 + 1], p[i * 3 + 2]); }

/* ray.h:66-78 */
static int hit_aabb(const float* bmin, const float* bmax, const ray_t* r) {
    v3 invdir = V3(1.0f / r->direction.x, 1.0f / r->direction.y, 1.0f / r->direction.z);
    v3 fv = mul3(sub3(V3(bmax[0], bmax[1], bmax[2]), r->origin), invdir);
    v3 nv = mul3(sub3(V3(bmin[0], bmin[1], bmin[2]), r->origin), invdir);
    v3 tmax = V3(fmax_(fv.x, nv.x), fmax_(fv.y, nv.y), fmax_(fv.z, nv.z));
    v3 tmin = V3(fmin_(fv.x, nv.x), fmin_(fv.y, nv.y), fmin_(fv.z, nv.z));
    float t1 = fmin_(fmin_(tmax.x, fmin_(tmax.y, tmax.z)), r->max_t);
    float t0 = fmax_(fmax_(fmax_(tmin.x, fmax_(tmin.y, tmin.z)), r->min_t), 0.0f);
    return t1 >= t0;
}

/* ray.h:137-163 on explicit corner positions */
static int hit_triangle_pos(v3 p0, v3 p1, v3 p2, const ray_t* r, float* t_out, v3* bary, v3* gn, int* nan_t) {
    v3 v2v1 = sub3(p1, p0);
    v3 v3v1 = sub3(p2, p0);
    v3 originv1 = sub3(r->origin, p0);
    v3 normal = cross3(v2v1, v3v1);
    v3 q = cross3(originv1, r->direction);
    float d = 1.0f / dot3(r->direction, normal);
    float u = d * dot3(neg3(q), v3v1);
    float v = d * dot3(q, v2v1);
    float t = d * dot3(neg3(normal), originv1);
    if (u < 0.0f || u > 1.0f || v < 0.0f || u + v > 1.0f || t < r->min_t || t > r->max_t) return 0;
    *t_out = t;
    *bary = V3(1.0f - u - v, u, v);
    *gn = normalize3(normal);
    if (nan_t) *nan_t = is_nan(t) || is_nan(u) || is_nan(v);
    return 1;
}
static int hit_triangle(const orc_frame* f, uint32_t id, const ray_t* r, hit_t* info, int* nan_t) {
    uint32_t idx[4];
    decode_triangle_indices(f, id, idx);
    int h = hit_triangle_pos(load3(f->positions, idx[0]), load3(f->positions, idx[1]), load3(f->positions, idx[2]), r,
                             &info->t, &info->barycentrics, &info->geometry_normal, nan_t);
    if (h) info->primitive_id = id;
    return h;
}

/* ray.h:105-135 (dead code in the reference: the call site is commented out, ray.h:173-177) */
static int hit_sphere(v3 center, float radius, const ray_t* r, float* t_out) {
    v3 oc = sub3(r->origin, center);
    float a = dot3(r->direction, r->direction);
    float half_b = dot3(r->direction, oc);
    float c = dot3(oc, oc) - radius * radius;
    float discriminant = half_b * half_b - a * c;
    if (discriminant < 0) return 0;
    float sd = sqrtf(discriminant);
    float root = (-half_b - sd) / a;
    if (root < r->min_t || root > r->max_t) {
        root = (-half_b + sd) / a;
        if (root < r->min_t || root > r->max_t) return 0;
    }
    *t_out = root;
    return 1;
}

/* ray.h:165-193.  `r` by value: max_t shrinks locally. */
static int hit_node(const orc_frame* f, ray_t r, hit_t* info, orc_counters* c, int* any_nan) {
    const node_t* nodes = (const node_t*)f->bvh_nodes;
    hit_t temp;
    int hit = 0;
    int32_t id = 0;
    uint64_t n_nodes = 0, n_tris = 0;
    while (id != -1) {
        const node_t* nd = &nodes[id];
        n_nodes++;
        if (nd->primitive_id != -1) {
            int nan_t = 0;
            n_tris++;
            if (hit_triangle(f, (uint32_t)nd->primitive_id, &r, &temp, &nan_t)) {
                *info = temp;
                r.max_t = temp.t;
                hit = 1;
                if (nan_t) { if (any_nan) *any_nan = 1; if (c) c->nan_hits++; }
            }
            id = nd->next_id;
        } else if (hit_aabb(nd->min, nd->max, &r)) {
            id++;
        } else {
            id = nd->next_id;
        }
    }
    if (c) { c->rays++; c->nodes += n_nodes; c->triangles += n_tris; }
    if (hit) info->point = add3(r.origin, scale3(r.direction, info->t)); /* ray.h:191 (only read after a hit) */
    return hit;
}

/* ray.h:80-103 */
static uint32_t hit_aabbs(const orc_frame* f, const ray_t* r) {
    const node_t* nodes = (const node_t*)f->bvh_nodes;
    uint32_t count = 0;
    int32_t id = 0;
    while (id != -1) {
        if (nodes[id].primitive_id != -1) id = nodes[id].next_id;
        else if (hit_aabb(nodes[id].min, nodes[id].max, r)) { count++; id++; }
        else id = nodes[id].next_id;
    }
    return count;
}

/* ------------------------------------------------------------------------------------------------
 * Closest hit over the flattened GPU layout.  This restates THIS REPOSITORY's traversal (not the
 * reference's): 4-wide nodes with 8-bit quantised child boxes, children visited nearest first through an
 * explicit stack, boxes padded by (1 + 2^-22) on the exit distance so they cull conservatively, leaves of
 * 1..4 pre-gathered triangles, equal-t ties resolved by the triangle's rank in the reference visiting
 * order (later-visited wins, as ray.h:153 does), NaN-t candidates rejected.  Used to count the nodes/triangles the GPU kernels fetch and to check
 * the flattened layout without a GPU.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t magic, version, node_count, triangle_count;
    uint64_t nodes_offset, triangles_offset;
    int32_t root; uint32_t max_depth;
    float scene_min[3], scene_max[3];
} flat_header_t;
typedef struct {
    float origin[3]; float step_x;
    int32_t child[4];
    uint32_t qlo[3]; uint32_t qhi_x;
    uint32_t qhi_y, qhi_z; float step_y, step_z;
} flat_node_t;
typedef struct {
    float p0[3]; int32_t rank;
    float e1[3]; int32_t primitive_id;
    float e2[3]; int32_t kind;   /* 0: triangle; 1: sphere -- p0 = centre, e1[0] = radius, primitive_id = FLAT_SPHERE_TAG | index */
    float n[3];  int32_t pad1;
} flat_tri_t;
#define FLAT_KIND_SPHERE 1
#define FLAT_SPHERE_TAG 0x40000000
typedef struct { float position[4]; uint8_t mat[48]; float radius; float pad[3]; } sphere_t; /* compute.comp:31-36, 80 bytes */

#define FLAT_PAD 1.0000002384185791015625f
#define FLAT_EMPTY_CHILD INT32_MIN

/* 4-wide node.  Per node and axis: A = step * (1/d), B = (origin - o) * (1/d); the distance to the child
 * plane with quantised offset q is t = fma(q, A, B).  The sign of 1/d selects which of a child's two planes
 * is the entry and which the exit.  NaN distances (0 * inf) are ignored by max/min = "no constraint".
 * Entry = max(entry planes, min_t), exit = min(exit planes, best_t), hit iff exit * (1 + 2^-22) >= entry.
 * The host pads the quantised boxes by 2^-20 of the node extent, which covers the rounding of A, B and the fma. */
static inline float max3_(float a, float b, float c) { return fmax_(fmax_(a, b), c); }
static inline float min3_(float a, float b, float c) { return fmin_(fmin_(a, b), c); }

typedef struct {
    float world_to_object[12];
    int32_t object_root;
    int32_t pad[3];
} flat_instance_t;
#define FLAT_INSTANCES_MAGIC 0x49525450u
#define FLAT_INSTANCE_TAG 0x40000000u

typedef struct {
    float best_t, best_u, best_v;
    int32_t best_rank, best_prim, best_inst;
    int hit;
    uint64_t n_nodes, n_tris;
} flat_state;

/* The ray below an instance: origin and direction through the rows of world_to_object, each component summed left to
 * right, the translation last; the direction keeps its length, so t means the same in both spaces. */
static inline v3 xform_point(const float* m, v3 p) {
    return V3(((m[0] * p.x + m[1] * p.y) + m[2] * p.z) + m[3], ((m[4] * p.x + m[5] * p.y) + m[6] * p.z) + m[7],
              ((m[8] * p.x + m[9] * p.y) + m[10] * p.z) + m[11]);
}
static inline v3 xform_dir(const float* m, v3 d) {
    return V3((m[0] * d.x + m[1] * d.y) + m[2] * d.z, (m[4] * d.x + m[5] * d.y) + m[6] * d.z, (m[8] * d.x + m[9] * d.y) + m[10] * d.z);
}
/* An object-space normal taken to world space: the transpose of world_to_object (not normalised here). */
static inline v3 xform_normal(const float* m, v3 n) {
    return V3((m[0] * n.x + m[4] * n.y) + m[8] * n.z, (m[1] * n.x + m[5] * n.y) + m[9] * n.z, (m[2] * n.x + m[6] * n.y) + m[10] * n.z);
}

/* One tree of the flattened layout, from reference `root`, for the ray (o, d) of instance `inst` (-1: no instancing).
 * A tree whose header carries the instances magic has instance leaves: each takes the ray into the object's space
 * and walks `object_blob` from the instance's root before the top level goes on -- the order in which the kernels'
 * single stack visits them.  Equal-t ties: the later (instance, rank) pair wins. */
static void flat_walk(const void* blob, int32_t root, v3 o, v3 d, float min_t, int32_t inst, const void* object_blob, flat_state* s) {
    const flat_header_t* h = (const flat_header_t*)blob;
    const flat_node_t* nodes = (const flat_node_t*)((const uint8_t*)blob + h->nodes_offset);
    const int top_level = h->magic == FLAT_INSTANCES_MAGIC;
    const flat_tri_t* tris = (const flat_tri_t*)((const uint8_t*)blob + h->triangles_offset);
    const flat_instance_t* instances = (const flat_instance_t*)((const uint8_t*)blob + h->triangles_offset);
    v3 inv = V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const int neg_x = is_neg(inv.x), neg_y = is_neg(inv.y), neg_z = is_neg(inv.z);
    int32_t stack[256];
    int sp = 0;
    int32_t cur = root;
    const float inf = INFINITY;
    for (;;) {
        if (cur >= 0) {
            const flat_node_t* nd = &nodes[cur];
            s->n_nodes++;
            const float Ax = nd->step_x * inv.x, Ay = nd->step_y * inv.y, Az = nd->step_z * inv.z;
            const float Bx = (nd->origin[0] - o.x) * inv.x, By = (nd->origin[1] - o.y) * inv.y, Bz = (nd->origin[2] - o.z) * inv.z;
            const uint32_t near_x = neg_x ? nd->qhi_x : nd->qlo[0], far_x = neg_x ? nd->qlo[0] : nd->qhi_x;
            const uint32_t near_y = neg_y ? nd->qhi_y : nd->qlo[1], far_y = neg_y ? nd->qlo[1] : nd->qhi_y;
            const uint32_t near_z = neg_z ? nd->qhi_z : nd->qlo[2], far_z = neg_z ? nd->qlo[2] : nd->qhi_z;
            float key[4];
            int32_t ref[4];
            for (int k = 0; k < 4; k++) {
                ref[k] = nd->child[k];
                key[k] = inf;
                if (ref[k] == FLAT_EMPTY_CHILD) continue;
                const int sh = 8 * k;
                const float nx = fmaf((float)((near_x >> sh) & 0xffu), Ax, Bx), fx = fmaf((float)((far_x >> sh) & 0xffu), Ax, Bx);
                const float ny = fmaf((float)((near_y >> sh) & 0xffu), Ay, By), fy = fmaf((float)((far_y >> sh) & 0xffu), Ay, By);
                const float nz = fmaf((float)((near_z >> sh) & 0xffu), Az, Bz), fz = fmaf((float)((far_z >> sh) & 0xffu), Az, Bz);
                const float t0 = fmax_(max3_(nx, ny, nz), min_t);
                const float t1 = fmin_(min3_(fx, fy, fz), s->best_t);
                if (t1 * FLAT_PAD >= t0) key[k] = t0;
            }
            /* three compare-exchanges (0,1)(2,3)(0,2): the nearest entry comes to slot 0, the farther of the two pair
             * winners -- most often the second nearest -- to slot 2, and slots 1 and 3 keep the loser of each pair.  (The
             * kernels' order; a full sort would save under 1 % of the node visits.  Misses carry +inf.) */
            static const int net[3][2] = { { 0, 1 }, { 2, 3 }, { 0, 2 } };
            for (int st = 0; st < 3; st++) {
                const int a = net[st][0], b = net[st][1];
                if (key[a] > key[b]) {
                    float tk = key[a]; key[a] = key[b]; key[b] = tk;
                    int32_t tr = ref[a]; ref[a] = ref[b]; ref[b] = tr;
                }
            }
            if (key[0] < inf) {
                /* the others that were hit go on the stack: slot 3, slot 1, then slot 2 on top */
                if (key[3] < inf) stack[sp++] = ref[3];
                if (key[1] < inf) stack[sp++] = ref[1];
                if (key[2] < inf) stack[sp++] = ref[2];
                cur = ref[0];
                continue;
            }
        } else if (top_level) {
            const uint32_t index = (uint32_t)~cur & ~FLAT_INSTANCE_TAG;
            const flat_instance_t* in = &instances[index];
            flat_walk(object_blob, in->object_root, xform_point(in->world_to_object, o), xform_dir(in->world_to_object, d), min_t,
                      (int32_t)index, NULL, s);
        } else {
            uint32_t ref = (uint32_t)~cur;
            uint32_t first = ref >> 2, count = (ref & 3u) + 1u;
            for (uint32_t k = 0; k < count; k++) {
                const flat_tri_t* tr = &tris[first + k];
                s->n_tris++;
                if (tr->kind == FLAT_KIND_SPHERE) {
                    /* a sphere leaf: hit_sphere (ray.h:105-135) against the ray shortened to the best hit so far, as the commented
                     * call site in hit_node would run it (ray.h:173-177); a NaN distance is no hit, ties go by rank like triangles */
                    ray_t sr; sr.origin = o; sr.direction = d; sr.min_t = min_t; sr.max_t = s->best_t;
                    float ts;
                    if (!hit_sphere(V3(tr->p0[0], tr->p0[1], tr->p0[2]), tr->e1[0], &sr, &ts) || is_nan(ts)) continue;
                    if (s->hit && ts == s->best_t && (inst < s->best_inst || (inst == s->best_inst && tr->rank < s->best_rank))) continue;
                    s->hit = 1; s->best_t = ts; s->best_u = 0.0f; s->best_v = 0.0f; s->best_rank = tr->rank; s->best_prim = tr->primitive_id;
                    s->best_inst = inst;
                    continue;
                }
                v3 e1 = V3(tr->e1[0], tr->e1[1], tr->e1[2]), e2 = V3(tr->e2[0], tr->e2[1], tr->e2[2]);
                v3 n = V3(tr->n[0], tr->n[1], tr->n[2]);
                v3 ov = sub3(o, V3(tr->p0[0], tr->p0[1], tr->p0[2]));
                v3 q = cross3(ov, d);
                float dd = 1.0f / dot3(d, n);
                float u = dd * dot3(neg3(q), e2);
                float v = dd * dot3(q, e1);
                float t = dd * dot3(neg3(n), ov);
                if (u < 0.0f || u > 1.0f || v < 0.0f || u + v > 1.0f || t < min_t || t > s->best_t) continue;
                if (is_nan(t) || is_nan(u) || is_nan(v)) continue;
                if (s->hit && t == s->best_t && (inst < s->best_inst || (inst == s->best_inst && tr->rank < s->best_rank))) continue;
                s->hit = 1; s->best_t = t; s->best_u = u; s->best_v = v; s->best_rank = tr->rank; s->best_prim = tr->primitive_id;
                s->best_inst = inst;
            }
        }
        if (sp == 0) break;
        cur = stack[--sp];
    }
}

/* Closest hit over a flattened tree, or over the instances in `instances_blob` of the tree in `blob` when that is
 * given; *inst_out (optional) receives the instance hit, -1 without instancing. */
int orc_closest_hit_two_level(const void* blob, const void* instances_blob, const float origin[3], const float dir[3], float min_t,
                              float max_t, int32_t* prim_out, int32_t* inst_out, float* t_out, float bary_out[3], orc_counters* c) {
    flat_state s;
    memset(&s, 0, sizeof s);
    s.best_t = max_t; s.best_rank = -1; s.best_prim = -1; s.best_inst = -1;
    const v3 o = V3(origin[0], origin[1], origin[2]), d = V3(dir[0], dir[1], dir[2]);
    if (instances_blob) flat_walk(instances_blob, ((const flat_header_t*)instances_blob)->root, o, d, min_t, -1, blob, &s);
    else flat_walk(blob, ((const flat_header_t*)blob)->root, o, d, min_t, -1, NULL, &s);
    if (c) { c->rays++; c->nodes += s.n_nodes; c->triangles += s.n_tris; }
    if (s.hit) {
        *prim_out = s.best_prim; *t_out = s.best_t;
        if (inst_out) *inst_out = s.best_inst;
        bary_out[0] = 1.0f - s.best_u - s.best_v; bary_out[1] = s.best_u; bary_out[2] = s.best_v;
    }
    return s.hit;
}

int orc_closest_hit_flat(const void* blob, const float origin[3], const float dir[3], float min_t, float max_t,
                         int32_t* prim_out, float* t_out, float bary_out[3], orc_counters* c) {
    return orc_closest_hit_two_level(blob, NULL, origin, dir, min_t, max_t, prim_out, NULL, t_out, bary_out, c);
}

static inline const float* instance_rows(const orc_frame* f, int32_t inst) {
    const flat_header_t* h = (const flat_header_t*)f->flat_instances;
    return ((const flat_instance_t*)((const uint8_t*)f->flat_instances + h->triangles_offset))[inst].world_to_object;
}

/* hit_node() semantics on top of the flattened layout: same outputs as hit_node (geometry normal from the
 * original vertex buffers, exactly as hit_triangle computes it). */
static int hit_node_flat(const orc_frame* f, ray_t r, hit_t* info, orc_counters* c) {
    int32_t prim; float t, bary[3];
    float o[3] = { r.origin.x, r.origin.y, r.origin.z }, d[3] = { r.direction.x, r.direction.y, r.direction.z };
    int32_t inst = -1;
    if (!orc_closest_hit_two_level(f->flat_bvh, f->flat_instances, o, d, r.min_t, r.max_t, &prim, &inst, &t, bary, c)) return 0;
    if (prim & FLAT_SPHERE_TAG) {
        /* the frame the reference's hit_sphere had in mind (its commented lines, ray.h:127-132): the point on the ray and the
         * outward normal (point - centre) / radius */
        const sphere_t* sp = &((const sphere_t*)f->spheres)[prim & ~FLAT_SPHERE_TAG];
        info->t = t;
        info->barycentrics = V3(bary[0], bary[1], bary[2]);
        info->point = add3(r.origin, scale3(r.direction, t));
        v3 out = sub3(info->point, V3(sp->position[0], sp->position[1], sp->position[2]));
        info->geometry_normal = V3(out.x / sp->radius, out.y / sp->radius, out.z / sp->radius);
        info->instance = inst;
        info->primitive_id = (uint32_t)prim;
        return 1;
    }
    uint32_t idx[4];
    decode_triangle_indices(f, (uint32_t)prim, idx);
    v3 p0 = load3(f->positions, idx[0]), p1 = load3(f->positions, idx[1]), p2 = load3(f->positions, idx[2]);
    info->t = t;
    info->barycentrics = V3(bary[0], bary[1], bary[2]);
    info->geometry_normal = normalize3(cross3(sub3(p1, p0), sub3(p2, p0)));
    info->instance = inst;
    if (inst >= 0) info->geometry_normal = normalize3(xform_normal(instance_rows(f, inst), info->geometry_normal));
    info->primitive_id = (uint32_t)prim;
    info->point = add3(r.origin, scale3(r.direction, t));
    return 1;
}

/* ------------------------------------------------------------------------------------------------
 * texture(sampler2D, uv): compute.comp:142-153 with the sampler of vk-api.cpp:406-424 (LINEAR, REPEAT, lod 0)
 * on RGBA8 UNORM images (gltf.cpp:126,151)
 * ---------------------------------------------------------------------------------------------- */
static inline float fract_repeat(float x) {
    float fr = x - floorf(x);
    if (!(fr == fr)) fr = 0.0f; /* NaN / inf coordinates sample texel 0 */
    return fr;
}
static v4 sample_texture(const orc_texture* tx, float u, float v) {
    v4 out = { 1.0f, 1.0f, 1.0f, 1.0f };
    if (!tx || !tx->rgba || tx->w == 0 || tx->h == 0) return out;
    const int32_t W = (int32_t)tx->w, H = (int32_t)tx->h;
    float x = fract_repeat(u) * (float)W - 0.5f;
    float y = fract_repeat(v) * (float)H - 0.5f;
    float fx = floorf(x), fy = floorf(y);
    float a = x - fx, b = y - fy;
    int32_t i0 = (int32_t)fx, j0 = (int32_t)fy;
    int32_t i1 = i0 + 1, j1 = j0 + 1;
    i0 = ((i0 % W) + W) % W; i1 = ((i1 % W) + W) % W;
    j0 = ((j0 % H) + H) % H; j1 = ((j1 % H) + H) % H;
    const uint8_t* t00 = tx->rgba + 4 * ((size_t)j0 * W + i0);
    const uint8_t* t10 = tx->rgba + 4 * ((size_t)j0 * W + i1);
    const uint8_t* t01 = tx->rgba + 4 * ((size_t)j1 * W + i0);
    const uint8_t* t11 = tx->rgba + 4 * ((size_t)j1 * W + i1);
    float o[4];
    for (int ch = 0; ch < 4; ch++) {
        float c00 = (float)t00[ch] / 255.0f, c10 = (float)t10[ch] / 255.0f;
        float c01 = (float)t01[ch] / 255.0f, c11 = (float)t11[ch] / 255.0f;
        float top = c00 * (1.0f - a) + c10 * a;
        float bot = c01 * (1.0f - a) + c11 * a;
        o[ch] = top * (1.0f - b) + bot * b;
    }
    out.x = o[0]; out.y = o[1]; out.z = o[2]; out.w = o[3];
    return out;
}
static v4 sample_bindless(const orc_frame* f, uint32_t texture_id, float u, float v) {
    if (texture_id == 0 || texture_id >= f->texture_count) { v4 w = { 1.0f, 1.0f, 1.0f, 1.0f }; return w; }
    return sample_texture(&f->textures[texture_id], u, v);
}

/* ------------------------------------------------------------------------------------------------
 * compute.comp:124-245
 * ---------------------------------------------------------------------------------------------- */
static uint32_t captured_primitive(const orc_frame* f, const hit_t* info) {
    if (!(f->use_flat && f->flat_instances) || info->instance < 0) return info->primitive_id;
    return (uint32_t)info->instance * ((const flat_header_t*)f->flat_bvh)->triangle_count + info->primitive_id;
}

#define BRDF_DIFFUSE 1u
#define BRDF_SPECULAR 2u

static v3 ray_color(const orc_frame* f, const meta_t* m, ray_t r, uint32_t seed, orc_counters* c, int32_t* aov_prim,
                    float* aov_t) {
    hit_t info;
    memset(&info, 0, sizeof info);
    v3 radiance = V3(0.0f, 0.0f, 0.0f);
    v3 throughput = V3(1.0f, 1.0f, 1.0f);
    const material_t* materials = (const material_t*)f->materials;

    for (uint32_t bounce = 0; bounce < m->max_bounce; bounce++) {
        int any_nan = 0;
        int hit = f->use_flat ? hit_node_flat(f, r, &info, c) : hit_node(f, r, &info, c, &any_nan);
        if (bounce < 2 && aov_prim) {
            aov_prim[bounce] = hit ? (any_nan ? -2 : (int32_t)captured_primitive(f, &info)) : -1;
            aov_t[bounce] = hit ? info.t : 0.0f;
        }
        if (!hit) return scale3(throughput, 1.0f);
        if (c) c->shaded++;

        v3 v = neg3(r.direction);

        v3 nrm[3];
        v2 tuv[3];
        const material_t* mat;
        if (f->use_flat && (info.primitive_id & FLAT_SPHERE_TAG)) {
            /* a sphere: the outward normal at all three "vertices", texture coordinates (0, 0), the material the sphere record embeds */
            nrm[0] = nrm[1] = nrm[2] = info.geometry_normal;
            tuv[0].x = tuv[0].y = tuv[1].x = tuv[1].y = tuv[2].x = tuv[2].y = 0.0f;
            mat = (const material_t*)((const sphere_t*)f->spheres)[info.primitive_id & ~FLAT_SPHERE_TAG].mat;
        } else {
            uint32_t idx[4];
            decode_triangle_indices(f, info.primitive_id, idx);
            nrm[0] = load3(f->normals, idx[0]); nrm[1] = load3(f->normals, idx[1]); nrm[2] = load3(f->normals, idx[2]);
            tuv[0].x = f->uvs[idx[0] * 2]; tuv[0].y = f->uvs[idx[0] * 2 + 1];
            tuv[1].x = f->uvs[idx[1] * 2]; tuv[1].y = f->uvs[idx[1] * 2 + 1];
            tuv[2].x = f->uvs[idx[2] * 2]; tuv[2].y = f->uvs[idx[2] * 2 + 1];
            mat = &materials[idx[3]];
        }

        v2 uv = interp2(tuv, info.barycentrics);
        v3 shading_normal = normalize3(interp3(nrm, info.barycentrics));
        /* below an instance the normals are the object's: world_to_object transposed takes them to world space */
        if (f->use_flat && f->flat_instances && info.instance >= 0)
            shading_normal = normalize3(xform_normal(instance_rows(f, info.instance), shading_normal));

        v4 albedo = sample_bindless(f, mat->albedo_tex, uv.x, uv.y);
        v3 diffuse_color = mul3(srgb_to_linear(V3(albedo.x, albedo.y, albedo.z)),
                                V3(mat->base_color[0], mat->base_color[1], mat->base_color[2]));
        v4 mrs = sample_bindless(f, mat->mr_tex, uv.x, uv.y);
        float metalness = mrs.x * mat->metalness;
        float roughness = mrs.y * mat->roughness;

        if (dot3(info.geometry_normal, v) < 0.f) info.geometry_normal = neg3(info.geometry_normal);
        if (dot3(info.geometry_normal, shading_normal) < 0.0f) shading_normal = neg3(shading_normal);

        uint32_t brdf_type;
        if (metalness == 1.f && roughness == 0.f) {
            brdf_type = BRDF_SPECULAR;
        } else {
            float p = get_brdf_probability(diffuse_color, metalness, neg3(r.direction), shading_normal);
            if (orc_rand(&seed) < p) {
                brdf_type = BRDF_SPECULAR;
                throughput = div3s(throughput, p);
            } else {
                brdf_type = BRDF_DIFFUSE;
                throughput = div3s(throughput, 1.0f - p);
            }
        }

        if (dot3(shading_normal, v) <= 0.f) break;

        v4 rotation_to_z = rotation_to_z_axis(shading_normal);
        v3 view_local = rotate_point(rotation_to_z, neg3(r.direction));
        v3 normal_local = V3(0.0f, 0.0f, 1.0f);
        v3 ray_dir_local = V3(0.0f, 0.0f, 0.0f);
        v3 sample_weight = V3(0.0f, 0.0f, 0.0f);

        if (brdf_type == BRDF_DIFFUSE) {
            float r0 = orc_rand(&seed);
            float r1 = orc_rand(&seed);
            ray_dir_local = sample_hemisphere(r0, r1);
            sample_weight = base_color_to_diffuse_reflectance(diffuse_color, metalness);
        } else {
            float alpha = roughness * roughness;
            float alpha_squared = alpha * alpha;
            v3 half_local;
            if (alpha == 0.0f) half_local = V3(0.0f, 0.0f, 1.0f);
            else half_local = sample_ggx_vndf(view_local, alpha, alpha, seed);

            /* reflect(I, N) = I - 2 * dot(N, I) * N with I = -view_local */
            v3 I = neg3(view_local);
            v3 light_local = sub3(I, scale3(half_local, 2.0f * dot3(half_local, I)));

            v3 specular_F0 = base_color_to_diffuse_reflectance(diffuse_color, metalness); /* sic: compute.comp:205 */

            float HdotL = clamp_(dot3(half_local, light_local), 0.00001f, 1.0f);
            float NdotL = clamp_(dot3(normal_local, light_local), 0.00001f, 1.0f);
            v3 fr = eval_fresnel_schlick(specular_F0, shadowed_f90(specular_F0), HdotL);
            float g = smith_g1_ggx(alpha, NdotL, alpha_squared, NdotL * NdotL);
            sample_weight = scale3(fr, g);
            ray_dir_local = light_local;
        }

        if (luminance(sample_weight) == 0.0f) break;

        v4 rotation_from_z = invert_rotation(rotation_to_z);
        v3 ray_dir = rotate_point(rotation_from_z, ray_dir_local);

        if (dot3(info.geometry_normal, ray_dir) <= 0.0f) break;

        throughput = mul3(throughput, sample_weight);

        r.origin = info.point; r.direction = ray_dir; r.min_t = 0.001f; r.max_t = 1e15f;

        if (bounce > m->min_bounce) {
            float probability = fmin_(0.95f, luminance(throughput));
            if (probability < orc_rand(&seed)) break;
            throughput = div3s(throughput, probability);
        }
    }
    return radiance;
}

/* compute.comp:247-262 */
static ray_t generate_camera_ray(const meta_t* m, int32_t cx, int32_t cy, uint32_t seed /* by value */) {
    float sx = (float)(m->width - 1u), sy = (float)(m->height - 1u);
    float uvx = (float)cx / sx, uvy = 1.0f - (float)cy / sy;
    float d0 = orc_rand(&seed);
    float d1 = orc_rand(&seed);
    v2 rand_disk = disk_vec(d0, d1);
    float j0 = orc_rand(&seed);
    float j1 = orc_rand(&seed);
    float jx = uvx + j0 / sx, jy = uvy + j1 / sy;
    v2 lens = { 0.0f, 0.0f };
    if (m->enable_dof == 1u) { lens.x = rand_disk.x * m->lens_radius; lens.y = rand_disk.y * m->lens_radius; }

    v3 right = V3(m->right[0], m->right[1], m->right[2]), up = V3(m->up[0], m->up[1], m->up[2]);
    v3 offset = add3(scale3(right, lens.x), scale3(up, lens.y));
    v3 fp = V3(m->first_pixel[0], m->first_pixel[1], m->first_pixel[2]);
    v3 hor = V3(m->horizontal[0], m->horizontal[1], m->horizontal[2]);
    v3 ver = V3(m->vertical[0], m->vertical[1], m->vertical[2]);
    v3 pos = V3(m->position[0], m->position[1], m->position[2]);
    v3 proj = add3(add3(fp, scale3(hor, jx)), scale3(ver, jy));
    ray_t r;
    r.origin = add3(pos, offset);
    r.direction = sub3(sub3(proj, pos), offset);
    r.min_t = 0.001f;
    r.max_t = 1e15f;
    return r;
}

static inline v4 image_load(const orc_image* im, int32_t x, int32_t y) {
    v4 z = { 0, 0, 0, 0 };
    if (x < 0 || y < 0 || (uint32_t)x >= im->w || (uint32_t)y >= im->h) return z;
    const float* p = im->rgba + 4 * ((size_t)y * im->w + (size_t)x);
    v4 r = { p[0], p[1], p[2], p[3] };
    return r;
}
static inline void image_store(const orc_image* im, int32_t x, int32_t y, v3 c) {
    if (x < 0 || y < 0 || (uint32_t)x >= im->w || (uint32_t)y >= im->h) return;
    float* p = im->rgba + 4 * ((size_t)y * im->w + (size_t)x);
    p[0] = c.x; p[1] = c.y; p[2] = c.z; p[3] = 1.0f;
}

/* compute.comp:264-289, one invocation */
static void shade_pixel(const orc_frame* f, uint32_t gx, uint32_t gy, orc_counters* c) {
    const meta_t* m = (const meta_t*)f->scene_meta;
    if (gx >= m->width || gy >= m->height) return;
    int32_t cx = (int32_t)gx * m->downscale_factor, cy = (int32_t)gy * m->downscale_factor;
    const uint32_t seed_index = f->seed_sample_index ? f->seed_sample_index : m->sample_index;
    uint32_t seed = ((uint32_t)cx * 1973u + (uint32_t)cy * 9277u + seed_index * 26699u) | 1u;
    ray_t r = generate_camera_ray(m, cx, cy, seed);

    int32_t* aov_prim = f->aov_prim ? f->aov_prim + 2 * ((size_t)gy * m->width + gx) : NULL;
    float* aov_t = f->aov_t ? f->aov_t + 2 * ((size_t)gy * m->width + gx) : NULL;
    if (aov_prim) { aov_prim[0] = aov_prim[1] = -1; aov_t[0] = aov_t[1] = 0.0f; }

    v3 out_color;
    if (m->debug_bvh == 1u) {
        float n = (float)hit_aabbs(f, &r);
        out_color = V3(n * 0.001f, n * 0.0f, n * 0.0f);
    } else {
        out_color = ray_color(f, m, r, seed, c, aov_prim, aov_t);
    }

    float spp_scale = 1.0f / (float)m->sample_index;
    v4 a = image_load(&f->accumulation, cx, cy);
    v3 acc = srgb_to_linear(V3(a.x, a.y, a.z));
    out_color = linear_to_srgb(V3(mix_(acc.x, out_color.x, spp_scale), mix_(acc.y, out_color.y, spp_scale),
                                  mix_(acc.z, out_color.z, spp_scale)));
    for (int32_t i = 0; i < m->downscale_factor; i++)
        for (int32_t j = 0; j < m->downscale_factor; j++) image_store(&f->output, cx + i, cy + j, out_color);
}

typedef struct {
    const orc_frame* f;
    uint32_t y1;
    volatile uint32_t* next_row;
    orc_counters counters;
} worker_t;

static void* render_worker(void* arg) {
    worker_t* w = (worker_t*)arg;
    const meta_t* m = (const meta_t*)w->f->scene_meta;
    for (;;) {
        uint32_t y = __atomic_fetch_add(w->next_row, 1u, __ATOMIC_RELAXED);
        if (y >= w->y1) break;
        for (uint32_t x = 0; x < m->width; x++) shade_pixel(w->f, x, y, &w->counters);
    }
    return NULL;
}

void orc_render(const orc_frame* f, uint32_t y0, uint32_t y1, int threads, orc_counters* counters) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    uint32_t next_row = y0;
    worker_t* w = (worker_t*)calloc((size_t)threads, sizeof(worker_t));
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    for (int i = 0; i < threads; i++) { w[i].f = f; w[i].y1 = y1; w[i].next_row = &next_row; }
    if (threads == 1) render_worker(&w[0]);
    else {
        for (int i = 0; i < threads; i++) pthread_create(&th[i], NULL, render_worker, &w[i]);
        for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
    }
    if (counters) {
        for (int i = 0; i < threads; i++) {
            counters->rays += w[i].counters.rays; counters->nodes += w[i].counters.nodes;
            counters->triangles += w[i].counters.triangles; counters->shaded += w[i].counters.shaded;
            counters->nan_hits += w[i].counters.nan_hits;
        }
    }
    free(w); free(th);
}

/* tonemapping.comp:19-30 (the shader has no bounds guard; out-of-image accesses follow the image rules above) */
void orc_tonemap(const orc_frame* f, uint32_t groups_x, uint32_t groups_y) {
    const meta_t* m = (const meta_t*)f->scene_meta;
    const int32_t ds = m->downscale_factor;
    for (uint32_t gy = 0; gy < groups_y * 8u; gy++) {
        for (uint32_t gx = 0; gx < groups_x * 8u; gx++) {
            int32_t cx = (int32_t)gx * ds, cy = (int32_t)gy * ds;
            v4 a = image_load(&f->accumulation, cx, cy);
            v3 o = V3(sqrtf(a.x), sqrtf(a.y), sqrtf(a.z));
            for (int32_t i = 0; i < ds; i++)
                for (int32_t j = 0; j < ds; j++) image_store(&f->output, cx + i, cy + j, o);
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * unit entry points
 * ---------------------------------------------------------------------------------------------- */
float orc_sin(float x) { float s, c; sincos_(x, &s, &c); return s; }
float orc_cos(float x) { float s, c; sincos_(x, &s, &c); return c; }
float orc_pow(float x, float y) { return pow_(x, y); }
void orc_srgb_to_linear(const float in[3], float out[3]) {
    v3 r = srgb_to_linear(V3(in[0], in[1], in[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_linear_to_srgb(const float in[3], float out[3]) {
    v3 r = linear_to_srgb(V3(in[0], in[1], in[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
static ray_t make_ray(const float o[3], const float d[3], float min_t, float max_t) {
    ray_t r; r.origin = V3(o[0], o[1], o[2]); r.direction = V3(d[0], d[1], d[2]); r.min_t = min_t; r.max_t = max_t;
    return r;
}
int orc_hit_aabb(const float bmin[3], const float bmax[3], const float origin[3], const float dir[3], float min_t, float max_t) {
    ray_t r = make_ray(origin, dir, min_t, max_t);
    return hit_aabb(bmin, bmax, &r);
}
int orc_hit_triangle_pos(const float p0[3], const float p1[3], const float p2[3], const float origin[3],
                         const float dir[3], float min_t, float max_t, float* t, float bary[3], float gn[3]) {
    ray_t r = make_ray(origin, dir, min_t, max_t);
    v3 b, g; float tt;
    if (!hit_triangle_pos(V3(p0[0], p0[1], p0[2]), V3(p1[0], p1[1], p1[2]), V3(p2[0], p2[1], p2[2]), &r, &tt, &b, &g, NULL)) return 0;
    *t = tt; bary[0] = b.x; bary[1] = b.y; bary[2] = b.z; gn[0] = g.x; gn[1] = g.y; gn[2] = g.z;
    return 1;
}
int orc_hit_sphere(const float center[3], float radius, const float origin[3], const float dir[3], float min_t,
                   float max_t, float* t) {
    ray_t r = make_ray(origin, dir, min_t, max_t);
    return hit_sphere(V3(center[0], center[1], center[2]), radius, &r, t);
}
void orc_rotation_to_z_axis(const float axis[3], float q[4]) {
    v4 r = rotation_to_z_axis(V3(axis[0], axis[1], axis[2])); q[0] = r.x; q[1] = r.y; q[2] = r.z; q[3] = r.w;
}
void orc_rotate_point(const float q[4], const float v[3], float out[3]) {
    v4 qq = { q[0], q[1], q[2], q[3] };
    v3 r = rotate_point(qq, V3(v[0], v[1], v[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_sample_hemisphere(float r0, float r1, float out[3]) { v3 r = sample_hemisphere(r0, r1); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void orc_disk_vec(float r0, float r1, float out[2]) { v2 r = disk_vec(r0, r1); out[0] = r.x; out[1] = r.y; }
void orc_sample_ggx_vndf(const float ve[3], float ax, float ay, uint32_t seed, float out[3]) {
    v3 r = sample_ggx_vndf(V3(ve[0], ve[1], ve[2]), ax, ay, seed); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float orc_brdf_probability(const float color[3], float metalness, const float view[3], const float n[3]) {
    return get_brdf_probability(V3(color[0], color[1], color[2]), metalness, V3(view[0], view[1], view[2]), V3(n[0], n[1], n[2]));
}
void orc_sample_texture(const orc_texture* t, float u, float v, float out[4]) {
    v4 r = sample_texture(t, u, v); out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
void orc_camera_ray(const void* scene_meta, int32_t cx, int32_t cy, uint32_t seed, float origin[3], float dir[3]) {
    ray_t r = generate_camera_ray((const meta_t*)scene_meta, cx, cy, seed);
    origin[0] = r.origin.x; origin[1] = r.origin.y; origin[2] = r.origin.z;
    dir[0] = r.direction.x; dir[1] = r.direction.y; dir[2] = r.direction.z;
}
int orc_closest_hit(const orc_frame* f, const float origin[3], const float dir[3], float min_t, float max_t,
                    int32_t* prim, float* t, float bary[3], orc_counters* c) {
    hit_t info;
    memset(&info, 0, sizeof info);
    if (!hit_node(f, make_ray(origin, dir, min_t, max_t), &info, c, NULL)) return 0;
    *prim = (int32_t)info.primitive_id; *t = info.t;
    bary[0] = info.barycentrics.x; bary[1] = info.barycentrics.y; bary[2] = info.barycentrics.z;
    return 1;
}
uint32_t orc_hit_aabbs(const orc_frame* f, const float origin[3], const float dir[3], float min_t, float max_t) {
    ray_t r = make_ray(origin, dir, min_t, max_t);
    return hit_aabbs(f, &r);
}
