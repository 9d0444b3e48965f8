// Device functions shared by the megakernel and the wavefront kernels: scene access, RNG, texture
// fetch, BRDF sampling, camera rays and temporal accumulation.  Each block names the reference shader
// lines whose behaviour it reproduces (shaders/compute.comp and shaders/include/*.h of the reference).
#pragma once
#pragma clang fp contract(off)

#include "ptr_device_math.h"

namespace ptr {

// ---- records the kernels read ---------------------------------------------------------------------
struct Material {  // 48 bytes, include/material.hpp:17-25 / compute.comp:21-29
    float base_color[4];
    uint32_t albedo_tex, albedo_sampler, mr_tex, mr_sampler;
    float metalness, roughness;
    float pad[2];
};

// compute.comp:31-36 (the leaf type of the reference's first scene; its call site is commented out, ray.h:173-177), 80 bytes
struct Sphere { float4 position; Material mat; float radius; float pad[3]; };
// a hit's primitive id with this bit set names sphere (id & ~tag) of SceneView::spheres; triangle ids stay below 2^29
constexpr int32_t kSpherePrimTag = 0x40000000;

struct ThreadedNode {  // 32 bytes, include/bvh.hpp:17-22 / compute.comp:60-65
    float min[3]; int32_t next_id;
    float max[3]; int32_t primitive_id;
};

struct SceneMeta {  // 160 bytes, compute.comp:45-79 (note: `up` before `right`, as the SHADER declares them)
    float position[4], forward[4], up[4], right[4], horizontal[4], vertical[4], first_pixel[4];
    float lens_radius, cam_pad[3];
    uint32_t max_bounce, min_bounce, width, height, sample_index, enable_dof, debug_bvh;
    int32_t downscale_factor;
};
static_assert(sizeof(SceneMeta) == 160, "scene metadata is 160 bytes");

struct ImageDesc {  // one entry of the bindless image table
    void* data;
    uint32_t width, height;
    uint32_t format;   // PTR_FORMAT_*
    uint32_t sampler;  // bindless sampler index
    // RGBA8 images the library owns: for every texel position (i, j) the four texels a bilinear REPEAT lookup with its
    // upper-left corner there reads -- (i, j), (i+1, j), (i, j+1), (i+1, j+1), the +1 wrapped -- as one 16-byte entry, the
    // entries in 2 x 2 tiles of 64 bytes (tile rows of ceil(width / 2) tiles).  A lookup is then ONE 16-byte load from one
    // 64-byte piece of memory instead of four 4-byte loads from up to four.  nullptr: no such copy, fetch from `data`.
    const uint32_t* footprints;
};
PTR_DEV size_t footprint_entry(uint32_t tiles_x, uint32_t i, uint32_t j) {
    return ((size_t)(j >> 1) * tiles_x + (i >> 1)) * 4u + ((j & 1u) << 1) + (i & 1u);
}

struct Capture { int32_t prim0; float t0; int32_t prim1; float t1; };

// Everything one "compute" dispatch needs; built on the host from the 64-byte constant block.
struct SceneView {
    SceneMeta meta;            // by value: the frame keeps the metadata it was dispatched with, whatever is written next
    const ThreadedNode* nodes;
    const uint32_t* indices;
    const float* positions;
    const float* normals;
    const float* uvs;
    const Material* materials;
    const ImageDesc* images;   // bindless table
    uint32_t image_count;
    uint32_t output_index, accumulation_index;
    Capture* capture;          // optional
    unsigned long long* stats; // optional: [rays, nodes, triangles, shaded]
    uint32_t part_rank, part_world, part_stripe_rows;  // tile partition: stripes of rows
    uint32_t sample_rank, sample_world;                 // sample partition: which global samples this GPU draws
    uint32_t index_words;      // 3: the reference's packed index words; 4: v0, v1, v2, material as plain words
    const float4* shade_records;  // optional: kShadeRecordVec4 float4 per triangle, see ShadeInputs / gather_shade_inputs
    uint32_t shade_record_count;
    uint32_t prev_image_plus_one;  // several samples per dispatch: image index + 1 that receives the result of the last sample but one (0: none)
    const float4* instance_records;  // optional (two-level scenes): 4 float4 per instance, rows of world-to-object first
    uint32_t instance_count;
    uint32_t object_triangles;       // two-level scenes: captured primitive ids read instance * object_triangles + primitive
    uint32_t batch_samples;          // wavefront pipeline: samples meta.sample_index, +1, ... traced by ONE dispatch (>= 1)
    const Sphere* spheres;           // optional: the records the flattened layout's sphere leaves name
    uint32_t sphere_count;
};

struct Ray { f3 origin, dir; float min_t, max_t; };
struct Hit { float t, u, v; int32_t prim; int32_t inst = -1; };  // prim < 0: miss; inst >= 0: the placed copy that was hit

// ---- placed copies: the ray below an instance, normals on the way back ------------------------------------------------
// Rows of world-to-object (3x3 part, translation), each component summed left to right, the translation last.  The
// direction keeps its length, so a distance t means the same in both spaces.
PTR_DEV f3 xform_point(float4 r0, float4 r1, float4 r2, f3 p) {
    return mk3(((r0.x * p.x + r0.y * p.y) + r0.z * p.z) + r0.w, ((r1.x * p.x + r1.y * p.y) + r1.z * p.z) + r1.w,
               ((r2.x * p.x + r2.y * p.y) + r2.z * p.z) + r2.w);
}
PTR_DEV f3 xform_dir(float4 r0, float4 r1, float4 r2, f3 d) {
    return mk3((r0.x * d.x + r0.y * d.y) + r0.z * d.z, (r1.x * d.x + r1.y * d.y) + r1.z * d.z, (r2.x * d.x + r2.y * d.y) + r2.z * d.z);
}
// object-space normal to world space: the transpose of world-to-object (not normalised here)
PTR_DEV f3 xform_normal(float4 r0, float4 r1, float4 r2, f3 n) {
    return mk3((r0.x * n.x + r1.x * n.y) + r2.x * n.z, (r0.y * n.x + r1.y * n.y) + r2.y * n.z, (r0.z * n.x + r1.z * n.y) + r2.z * n.z);
}

// ---- rand.h:1-12 -------------------------------------------------------------------------------------
PTR_DEV uint32_t pcg_hash(uint32_t& seed) {
    const uint32_t state = seed;
    seed = state * (747796405u + 2891336453u);  // the shader's `seed *= a + b`: multiplier a+b, no increment
    const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
PTR_DEV float rand01(uint32_t& seed) { return (float)pcg_hash(seed) / 4294967296.0f; }

// ---- color_utils.h:1-26 -----------------------------------------------------------------------------
PTR_DEV float linear_to_srgb1(float c) {
    c = clampf(c, 0.0f, 1.0f);
    return mixf(pow_pos(c, 1.0f / 2.4f) * 1.055f - 0.055f, c * 12.92f, c < 0.0031308f ? 1.0f : 0.0f);
}
PTR_DEV float srgb_to_linear1(float c) {
    c = clampf(c, 0.0f, 1.0f);
    return mixf(pow_pos((c + 0.055f) / 1.055f, 2.4f), c / 12.92f, c < 0.04045f ? 1.0f : 0.0f);
}
PTR_DEV f3 linear_to_srgb(f3 c) { return mk3(linear_to_srgb1(c.x), linear_to_srgb1(c.y), linear_to_srgb1(c.z)); }
PTR_DEV f3 srgb_to_linear(f3 c) { return mk3(srgb_to_linear1(c.x), srgb_to_linear1(c.y), srgb_to_linear1(c.z)); }
PTR_DEV float luminance(f3 c) { return dot(c, mk3(0.2126f, 0.7152f, 0.0722f)); }

// ---- math.h:12-45 ------------------------------------------------------------------------------------
constexpr float kPi = 3.141592653589793f;
constexpr float kTwoPi = kPi * 2.0f;

PTR_DEV f3 sample_hemisphere(float r0, float r1) {
    const float radius = sqrtf(r0);
    const float theta = 2.0f * kPi * r1;
    float s, c;
    sincos_poly(theta, s, c);
    return mk3(c * radius, s * radius, sqrtf(1.0f - r0));
}
PTR_DEV f2 disk_vec(float r0, float r1) {
    const float z = r0 * 2.0f - 1.0f;
    const float a = 2.0f * kPi * r1;
    const float r = sqrtf(1.0f - z * z);
    float s, c;
    sincos_poly(a, s, c);
    f2 o; o.x = c * r; o.y = s * r;
    return o;
}
PTR_DEV f4 rotation_to_z_axis(f3 axis) {
    f4 q;
    if (axis.z < -0.99999f) { q.x = 1.0f; q.y = 0.0f; q.z = 0.0f; q.w = 0.0f; return q; }
    const float x = axis.y, y = -axis.x, z = 0.0f, w = 1.0f + axis.z;
    const float inv = rcp(sqrtf(x * x + y * y + z * z + w * w));
    q.x = x * inv; q.y = y * inv; q.z = z * inv; q.w = w * inv;
    return q;
}
PTR_DEV f4 invert_rotation(f4 q) { f4 r; r.x = -q.x; r.y = -q.y; r.z = -q.z; r.w = q.w; return r; }
PTR_DEV f3 rotate_point(f4 q, f3 v) {
    const f3 qa = mk3(q.x, q.y, q.z);
    const f3 a = qa * (2.0f * dot(qa, v));
    const f3 b = v * (q.w * q.w - dot(qa, qa));
    const f3 c = cross(qa, v) * (2.0f * q.w);
    return (a + b) + c;
}

// ---- brdf.h:13-99 --------------------------------------------------------------------------------------
constexpr float kMinDielectricsF0 = 0.04f;

// `seed` by value: the two draws made here do not advance the caller's generator (brdf.h:13).
PTR_DEV f3 sample_ggx_vndf(f3 Ve, float ax, float ay, uint32_t seed) {
    const f3 Vh = normalize(mk3(ax * Ve.x, ay * Ve.y, Ve.z));
    const float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    const f3 T1 = lensq > 0.0f ? mk3(-Vh.y, Vh.x, 0.0f) * rcp(sqrtf(lensq)) : mk3(1.0f, 0.0f, 0.0f);
    const f3 T2 = cross(Vh, T1);
    const float r = sqrtf(rand01(seed));
    const float phi = kTwoPi * rand01(seed);
    float sphi, cphi;
    sincos_poly(phi, sphi, cphi);
    const float t1 = r * cphi;
    float t2 = r * sphi;
    const float s = 0.5f * (1.0f + Vh.z);
    t2 = mixf(sqrtf(1.0f - t1 * t1), t2, s);
    const f3 Nh = ((T1 * t1) + (T2 * t2)) + (Vh * sqrtf(max_num(0.0f, 1.0f - t1 * t1 - t2 * t2)));
    return normalize(mk3(ax * Nh.x, ay * Nh.y, max_num(0.0f, Nh.z)));
}
PTR_DEV float smith_g1_ggx(float alpha_squared, float NdotS_squared) {
    return 2.0f / (sqrtf(((alpha_squared * (1.0f - NdotS_squared)) + NdotS_squared) / NdotS_squared) + 1.0f);
}
PTR_DEV f3 diffuse_reflectance(f3 base, float metalness) { return base * (1.0f - metalness); }
PTR_DEV f3 fresnel_schlick(f3 f0, float f90, float NdotS) {
    const float p = pow5(1.0f - NdotS);
    return mk3(f0.x + (f90 - f0.x) * p, f0.y + (f90 - f0.y) * p, f0.z + (f90 - f0.z) * p);
}
PTR_DEV float shadowed_f90(f3 F0) { return min_num(1.0f, (1.0f / kMinDielectricsF0) * luminance(F0)); }
PTR_DEV float brdf_probability(f3 color, float metalness, f3 view, f3 shading_normal) {
    const float l = luminance(mk3(mixf(kMinDielectricsF0, color.x, metalness), mixf(kMinDielectricsF0, color.y, metalness),
                                  mixf(kMinDielectricsF0, color.z, metalness)));
    const f3 specular_F0 = mk3(l, l, l);
    const float diffuse_refl = luminance(diffuse_reflectance(color, metalness));
    const float fresnel =
        clampf(luminance(fresnel_schlick(specular_F0, shadowed_f90(specular_F0), max_num(0.0f, dot(view, shading_normal)))), 0.0f, 1.0f);
    const float specular = fresnel;
    const float diffuse = diffuse_refl * (1.0f - fresnel);
    const float p = specular / max_num(0.0001f, specular + diffuse);
    return clampf(p, 0.1f, 0.9f);
}

// ---- texture(sampler2D, uv): compute.comp:142-153; LINEAR / REPEAT / lod 0 on RGBA8 UNORM ------------------
PTR_DEV float fract_repeat(float x) {
    float fr = x - floorf(x);
    if (!(fr == fr)) fr = 0.0f;
    return fr;
}
// b / 255 for a byte b, correctly rounded, without the division sequence: one multiply by RN(1/255) and one
// residual correction (f' = f + (b - 255 f) / 255, both steps fused).  Equal to the IEEE quotient for all 256
// inputs (checked exhaustively in tests/test_device_identities.py), at 3 instructions instead of ~12.
PTR_DEV float unorm8(uint32_t b) {
    const float k = 1.0f / 255.0f;
    const float fb = (float)b;
    const float f = fb * k;
    const float r = __builtin_fmaf(-f, 255.0f, fb);
    return __builtin_fmaf(r, k, f);
}
PTR_DEV f4 unpack_rgba8(uint32_t p) {
    f4 r;
    r.x = unorm8(p & 0xffu);
    r.y = unorm8((p >> 8) & 0xffu);
    r.z = unorm8((p >> 16) & 0xffu);
    r.w = unorm8(p >> 24);
    return r;
}
PTR_DEV f4 sample_bindless(const SceneView& sv, uint32_t texture_id, float u, float v) {
    f4 white; white.x = white.y = white.z = white.w = 1.0f;
    if (texture_id == 0u || texture_id >= sv.image_count) return white;
    const ImageDesc d = sv.images[texture_id];
    if (d.data == nullptr || d.width == 0u || d.height == 0u) return white;
    const int32_t W = (int32_t)d.width, H = (int32_t)d.height;
    const float x = fract_repeat(u) * (float)W - 0.5f;
    const float y = fract_repeat(v) * (float)H - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const float a = x - fx, b = y - fy;
    int32_t i0 = (int32_t)fx, j0 = (int32_t)fy;
    int32_t i1 = i0 + 1, j1 = j0 + 1;
    // x in [-0.5, W - 0.5]: one conditional wrap replaces the modulo
    i0 = i0 < 0 ? i0 + W : (i0 >= W ? i0 - W : i0);
    i1 = i1 < 0 ? i1 + W : (i1 >= W ? i1 - W : i1);
    j0 = j0 < 0 ? j0 + H : (j0 >= H ? j0 - H : j0);
    j1 = j1 < 0 ? j1 + H : (j1 >= H ? j1 - H : j1);
    uint32_t p00, p10, p01, p11;
    if (d.footprints) {
        const uint4 e = ((const uint4*)d.footprints)[footprint_entry((d.width + 1u) >> 1, (uint32_t)i0, (uint32_t)j0)];
        p00 = e.x; p10 = e.y; p01 = e.z; p11 = e.w;
    } else {
        const uint32_t* texels = (const uint32_t*)d.data;
        p00 = texels[(size_t)j0 * W + i0]; p10 = texels[(size_t)j0 * W + i1];
        p01 = texels[(size_t)j1 * W + i0]; p11 = texels[(size_t)j1 * W + i1];
    }
    const f4 c00 = unpack_rgba8(p00);
    const f4 c10 = unpack_rgba8(p10);
    const f4 c01 = unpack_rgba8(p01);
    const f4 c11 = unpack_rgba8(p11);
    f4 o;
    o.x = (c00.x * (1.0f - a) + c10.x * a) * (1.0f - b) + (c01.x * (1.0f - a) + c11.x * a) * b;
    o.y = (c00.y * (1.0f - a) + c10.y * a) * (1.0f - b) + (c01.y * (1.0f - a) + c11.y * a) * b;
    o.z = (c00.z * (1.0f - a) + c10.z * a) * (1.0f - b) + (c01.z * (1.0f - a) + c11.z * a) * b;
    o.w = (c00.w * (1.0f - a) + c10.w * a) * (1.0f - b) + (c01.w * (1.0f - a) + c11.w * a) * b;
    return o;
}

// ---- ray.h:20-54 ----------------------------------------------------------------------------------------
PTR_DEV void decode_triangle(const SceneView& sv, uint32_t tri, uint32_t& i0, uint32_t& i1, uint32_t& i2, uint32_t& mat) {
    if (sv.index_words == 4u) {  // extension for scenes past 2^24 vertices: four plain words per triangle
        const uint4 w = ((const uint4*)sv.indices)[tri];
        i0 = w.x; i1 = w.y; i2 = w.z; mat = w.w;
        return;
    }
    const uint32_t a = sv.indices[tri * 3u], b = sv.indices[tri * 3u + 1u], c = sv.indices[tri * 3u + 2u];
    i0 = a & 0x00ffffffu; i1 = b & 0x00ffffffu; i2 = c & 0x00ffffffu;
    mat = ((a & 0xff000000u) >> 8) | ((b & 0xff000000u) >> 16) | ((c & 0xff000000u) >> 24);
}
PTR_DEV f3 load3(const float* p, uint32_t i) { return mk3(p[i * 3u], p[i * 3u + 1u], p[i * 3u + 2u]); }

// ---- compute.comp:247-262 -----------------------------------------------------------------------------
PTR_DEV Ray generate_camera_ray(const SceneMeta& m, int32_t cx, int32_t cy, uint32_t seed /* by value */) {
    const float sx = (float)(m.width - 1u), sy = (float)(m.height - 1u);
    const float uvx = (float)cx / sx, uvy = 1.0f - (float)cy / sy;
    const float d0 = rand01(seed);
    const float d1 = rand01(seed);
    const f2 rand_disk = disk_vec(d0, d1);
    const float j0 = rand01(seed);
    const float j1 = rand01(seed);
    const float jx = uvx + j0 / sx, jy = uvy + j1 / sy;
    float lx = 0.0f, ly = 0.0f;
    if (m.enable_dof == 1u) { lx = rand_disk.x * m.lens_radius; ly = rand_disk.y * m.lens_radius; }
    const f3 offset = (mk3(m.right[0], m.right[1], m.right[2]) * lx) + (mk3(m.up[0], m.up[1], m.up[2]) * ly);
    const f3 pos = mk3(m.position[0], m.position[1], m.position[2]);
    const f3 proj = (mk3(m.first_pixel[0], m.first_pixel[1], m.first_pixel[2]) +
                     (mk3(m.horizontal[0], m.horizontal[1], m.horizontal[2]) * jx)) +
                    (mk3(m.vertical[0], m.vertical[1], m.vertical[2]) * jy);
    Ray r;
    r.origin = pos + offset;
    r.dir = (proj - pos) - offset;
    r.min_t = 0.001f;
    r.max_t = 1e15f;
    return r;
}

PTR_DEV uint32_t pixel_seed(int32_t cx, int32_t cy, uint32_t sample_index) {  // compute.comp:270
    return ((uint32_t)cx * 1973u + (uint32_t)cy * 9277u + sample_index * 26699u) | 1u;
}

// What the shade step needs of the hit triangle (ray.h:32-64, 146): vertex normals, uvs, material id, geometry normal.
struct ShadeInputs { f3 n0, n1, n2; f2 uv0, uv1, uv2; uint32_t material; f3 geometry_normal; };

// Fetched the shader's way: three index words, then nine scattered vertex records.
PTR_DEV ShadeInputs gather_shade_inputs(const SceneView& sv, uint32_t tri) {
    ShadeInputs s;
    uint32_t i0, i1, i2;
    decode_triangle(sv, tri, i0, i1, i2, s.material);
    const f3 p0 = load3(sv.positions, i0), p1 = load3(sv.positions, i1), p2 = load3(sv.positions, i2);
    s.geometry_normal = normalize(cross(p1 - p0, p2 - p0));
    s.n0 = load3(sv.normals, i0); s.n1 = load3(sv.normals, i1); s.n2 = load3(sv.normals, i2);
    s.uv0.x = sv.uvs[i0 * 2u]; s.uv0.y = sv.uvs[i0 * 2u + 1u];
    s.uv1.x = sv.uvs[i1 * 2u]; s.uv1.y = sv.uvs[i1 * 2u + 1u];
    s.uv2.x = sv.uvs[i2 * 2u]; s.uv2.y = sv.uvs[i2 * 2u + 1u];
    return s;
}

// The same values from one contiguous 80-byte record per triangle (built once per scene from the arrays above by
// ptr_wf_build_shade_records): 5 wide loads instead of 30 scattered ones.
constexpr uint32_t kShadeRecordVec4 = 5;
PTR_DEV void store_shade_record(float4* rec, const ShadeInputs& s) {
    rec[0] = make_float4(s.n0.x, s.n0.y, s.n0.z, s.n1.x);
    rec[1] = make_float4(s.n1.y, s.n1.z, s.n2.x, s.n2.y);
    rec[2] = make_float4(s.n2.z, s.uv0.x, s.uv0.y, s.uv1.x);
    rec[3] = make_float4(s.uv1.y, s.uv2.x, s.uv2.y, __uint_as_float(s.material));
    rec[4] = make_float4(s.geometry_normal.x, s.geometry_normal.y, s.geometry_normal.z, 0.0f);
}
PTR_DEV ShadeInputs load_shade_record(const float4* rec) {
    const float4 a = rec[0], b = rec[1], c = rec[2], d = rec[3], e = rec[4];
    ShadeInputs s;
    s.n0 = mk3(a.x, a.y, a.z); s.n1 = mk3(a.w, b.x, b.y); s.n2 = mk3(b.z, b.w, c.x);
    s.uv0.x = c.y; s.uv0.y = c.z; s.uv1.x = c.w; s.uv1.y = d.x; s.uv2.x = d.y; s.uv2.y = d.z;
    s.material = __float_as_uint(d.w);
    s.geometry_normal = mk3(e.x, e.y, e.z);
    return s;
}

// ---- one iteration of the bounce loop after the closest-hit query: compute.comp:130-241 -----------------
// Inputs: the ray that was traced, its closest hit, the path state.  Returns true when the path goes on
// (r holds the next ray), false when it ended with `result` (throughput on a miss, black otherwise).
PTR_DEV bool shade_bounce(const SceneView& sv, uint32_t bounce, uint32_t min_bounce, const Hit& hit, Ray& r, f3& throughput,
                          uint32_t& seed, f3& result) {
    if (hit.prim < 0) { result = throughput * 1.0f; return false; }
    result = mk3(0.0f, 0.0f, 0.0f);

    const f3 v = -r.dir;
    const f3 bary = mk3(1.0f - hit.u - hit.v, hit.u, hit.v);
    ShadeInputs in;
    const f3 point = r.origin + r.dir * hit.t;   // ray.h:191
    const bool on_sphere = (hit.prim & kSpherePrimTag) != 0;
    Material sphere_material;
    if (on_sphere) {
        // a sphere leaf: the frame hit_sphere's commented lines describe (ray.h:127-132) -- the outward normal
        // (point - centre) / radius, here at all three "vertices" -- texture coordinates (0, 0), the record's own material
        const uint32_t index = (uint32_t)(hit.prim & ~kSpherePrimTag);
        if (!sv.spheres || index >= sv.sphere_count) return false;
        const Sphere s = sv.spheres[index];
        const f3 out = point - mk3(s.position.x, s.position.y, s.position.z);
        in.geometry_normal = mk3(out.x / s.radius, out.y / s.radius, out.z / s.radius);
        in.n0 = in.n1 = in.n2 = in.geometry_normal;
        in.uv0.x = in.uv0.y = in.uv1.x = in.uv1.y = in.uv2.x = in.uv2.y = 0.0f;
        in.material = 0u;
        sphere_material = s.mat;
    } else if (sv.shade_records) {
        if ((uint32_t)hit.prim >= sv.shade_record_count) return false;  // a hit that names no triangle of the index buffer
        in = load_shade_record(sv.shade_records + (size_t)hit.prim * kShadeRecordVec4);
    } else {
        in = gather_shade_inputs(sv, (uint32_t)hit.prim);
    }

    // geometry normal as hit_triangle produces it (ray.h:146,159)
    f3 geometry_normal = in.geometry_normal;

    f3 shading_normal = normalize(((in.n0 * bary.x) + (in.n1 * bary.y)) + (in.n2 * bary.z));
    if (sv.instance_records && hit.inst >= 0) {
        if ((uint32_t)hit.inst >= sv.instance_count) return false;
        const float4* rec = sv.instance_records + 4 * (size_t)hit.inst;
        const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
        geometry_normal = normalize(xform_normal(r0, r1, r2, geometry_normal));
        shading_normal = normalize(xform_normal(r0, r1, r2, shading_normal));
    }
    const float tu = (in.uv0.x * bary.x + in.uv1.x * bary.y) + in.uv2.x * bary.z;
    const float tv = (in.uv0.y * bary.x + in.uv1.y * bary.y) + in.uv2.y * bary.z;

    const Material mat = on_sphere ? sphere_material : sv.materials[in.material];
    const f4 albedo = sample_bindless(sv, mat.albedo_tex, tu, tv);
    const f3 diffuse_color = srgb_to_linear(mk3(albedo.x, albedo.y, albedo.z)) * mk3(mat.base_color[0], mat.base_color[1], mat.base_color[2]);
    const f4 mr = sample_bindless(sv, mat.mr_tex, tu, tv);
    const float metalness = mr.x * mat.metalness;  // R as metalness, G as roughness: the reference's choice
    const float roughness = mr.y * mat.roughness;

    if (dot(geometry_normal, v) < 0.f) geometry_normal = -geometry_normal;
    if (dot(geometry_normal, shading_normal) < 0.0f) shading_normal = -shading_normal;

    bool specular;
    if (metalness == 1.f && roughness == 0.f) {
        specular = true;
    } else {
        const float p = brdf_probability(diffuse_color, metalness, -r.dir, shading_normal);
        if (rand01(seed) < p) { specular = true; throughput = throughput / p; }
        else { specular = false; throughput = throughput / (1.0f - p); }
    }

    if (dot(shading_normal, v) <= 0.f) return false;

    const f4 to_z = rotation_to_z_axis(shading_normal);
    const f3 view_local = rotate_point(to_z, -r.dir);
    f3 dir_local, weight;
    if (!specular) {
        const float r0 = rand01(seed);
        const float r1 = rand01(seed);
        dir_local = sample_hemisphere(r0, r1);
        weight = diffuse_reflectance(diffuse_color, metalness);
    } else {
        const float alpha = roughness * roughness;
        const float alpha_squared = alpha * alpha;
        const f3 half_local = (alpha == 0.0f) ? mk3(0.0f, 0.0f, 1.0f) : sample_ggx_vndf(view_local, alpha, alpha, seed);
        const f3 incident = -view_local;
        const f3 light_local = incident - half_local * (2.0f * dot(half_local, incident));  // reflect(I, N)
        const f3 specular_F0 = diffuse_reflectance(diffuse_color, metalness);                // as compute.comp:205 has it
        const float HdotL = clampf(dot(half_local, light_local), 0.00001f, 1.0f);
        const float NdotL = clampf(dot(mk3(0.0f, 0.0f, 1.0f), light_local), 0.00001f, 1.0f);
        const f3 fr = fresnel_schlick(specular_F0, shadowed_f90(specular_F0), HdotL);
        weight = fr * smith_g1_ggx(alpha_squared, NdotL * NdotL);
        dir_local = light_local;
    }

    if (luminance(weight) == 0.0f) return false;

    const f3 ray_dir = rotate_point(invert_rotation(to_z), dir_local);
    if (dot(geometry_normal, ray_dir) <= 0.0f) return false;

    throughput = throughput * weight;
    r.origin = point; r.dir = ray_dir; r.min_t = 0.001f; r.max_t = 1e15f;

    if (bounce > min_bounce) {  // Russian roulette
        const float probability = min_num(0.95f, luminance(throughput));
        if (probability < rand01(seed)) return false;
        throughput = throughput / probability;
    }
    return true;
}

// ---- image access + temporal accumulation: compute.comp:280-288 ----------------------------------------
PTR_DEV f4 image_load(const ImageDesc& im, int32_t x, int32_t y) {
    f4 z; z.x = z.y = z.z = z.w = 0.0f;
    if (im.data == nullptr || x < 0 || y < 0 || (uint32_t)x >= im.width || (uint32_t)y >= im.height) return z;
    const float4 t = ((const float4*)im.data)[(size_t)y * im.width + (size_t)x];
    f4 r; r.x = t.x; r.y = t.y; r.z = t.z; r.w = t.w;
    return r;
}
PTR_DEV void image_store(const ImageDesc& im, int32_t x, int32_t y, f3 c) {
    if (im.data == nullptr || x < 0 || y < 0 || (uint32_t)x >= im.width || (uint32_t)y >= im.height) return;
    ((float4*)im.data)[(size_t)y * im.width + (size_t)x] = make_float4(c.x, c.y, c.z, 1.0f);
}
// The accumulate step of `count` consecutive frames on one pixel, samples[k * stride] being frame k's sample: what `count`
// dispatches with the host's image swap in between (src/main.cpp:416) leave in the image written last.  Each frame decodes
// what the frame before stored -- the encode / decode round trip is not the identity in fp32, so it is done, not skipped --
// and a pixel outside the images reads 0 every time, as image_load gives it.  Both images must have the same extent.
// `before_last` (optional) receives what the frame before the last one stored -- the image the last frame read -- so that BOTH
// images end up as `count` single dispatches leave them (coalesced dispatches, "dispatch.coalesce").
PTR_DEV void accumulate_pixel_chain(const ImageDesc& acc, const ImageDesc& out, const ImageDesc* before_last, int32_t cx, int32_t cy, int32_t downscale,
                                    uint32_t first_sample_index, uint32_t count, const float4* samples, size_t stride) {
    const f4 a = image_load(acc, cx, cy);
    f3 stored = mk3(a.x, a.y, a.z);
    const bool inside = out.data != nullptr && cx >= 0 && cy >= 0 && (uint32_t)cx < out.width && (uint32_t)cy < out.height;
    f3 o = stored, previous = stored;
    for (uint32_t k = 0; k < count; k++) {
        const float4 s4 = samples[(size_t)k * stride];
        const float spp_scale = 1.0f / (float)(first_sample_index + k);
        const f3 prev = srgb_to_linear(stored);
        previous = o;
        o = linear_to_srgb(mk3(mixf(prev.x, s4.x, spp_scale), mixf(prev.y, s4.y, spp_scale), mixf(prev.z, s4.z, spp_scale)));
        stored = inside ? o : mk3(0.0f, 0.0f, 0.0f);
    }
    for (int32_t i = 0; i < downscale; i++)
        for (int32_t j = 0; j < downscale; j++) {
            // `before_last` may be the image `acc` was read from: this thread read its texel above and is the only one to write it
            if (before_last != nullptr && count >= 2u) image_store(*before_last, cx + i, cy + j, previous);
            image_store(out, cx + i, cy + j, o);
        }
}

PTR_DEV void accumulate_pixel(const ImageDesc& acc, const ImageDesc& out, int32_t cx, int32_t cy, int32_t downscale,
                              uint32_t sample_index, f3 sample) {
    const float spp_scale = 1.0f / (float)sample_index;
    const f4 a = image_load(acc, cx, cy);
    const f3 prev = srgb_to_linear(mk3(a.x, a.y, a.z));
    const f3 o = linear_to_srgb(mk3(mixf(prev.x, sample.x, spp_scale), mixf(prev.y, sample.y, spp_scale), mixf(prev.z, sample.z, spp_scale)));
    for (int32_t i = 0; i < downscale; i++)
        for (int32_t j = 0; j < downscale; j++) image_store(out, cx + i, cy + j, o);
}

// Sample partition: frame `sample_index` of GPU `sample_rank` draws the global sample
// (sample_index - 1) * world + rank + 1, so the GPUs of a node draw disjoint sample sets; the temporal
// weight stays 1 / sample_index (each GPU keeps the running mean of its own samples).
PTR_DEV uint32_t seed_sample_index(const SceneView& sv, uint32_t sample_index) {
    return sv.sample_world <= 1u ? sample_index : (sample_index - 1u) * sv.sample_world + sv.sample_rank + 1u;
}

// Does this context own image row `gy` of the dispatch under the stripe partition?
PTR_DEV bool owns_row(const SceneView& sv, uint32_t gy) {
    return sv.part_world <= 1u || ((gy / sv.part_stripe_rows) % sv.part_world) == sv.part_rank;
}

}  // namespace ptr
