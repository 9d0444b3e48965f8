/* TEST INFRASTRUCTURE ONLY -- the parity oracle.  Never linked into, or called from, the product.
 *
 * CPU restatement (plain C, fp32, no FMA contraction) of the reference's GPU path:
 *   shaders/compute.comp, shaders/include/{ray,brdf,math,rand,color_utils}.h, shaders/tonemapping.comp
 * Each function in oracle.c cites the reference lines it follows.
 *
 * PARITY STATUS: the reference holds no tests, golden images or known-answer vectors for this path and
 * its GLSL cannot be compiled or run in the build container (no glslc / Vulkan / lavapipe), so the
 * shader arithmetic here is "parity unpinned" by the reference itself: it is pinned by this restatement
 * plus hand-derived known-answer tests (tests/test_oracle_kat.py).  The host-side producers of its inputs
 * (BVH node array, camera block) ARE pinned: by golden fixtures generated from the unmodified reference
 * C++ (oracle/gen_golden.py).
 *
 * Arithmetic contract (what GLSL leaves implementation-defined is fixed here, and the HIP kernels
 * implement the same contract independently):
 *   - fp32 IEEE-754 round-to-nearest-even for + - * / sqrt, no contraction, denormals kept;
 *   - min/max = IEEE minimumNumber/maximumNumber (a NaN operand is ignored, -0 < +0);
 *   - normalize(v) = v * (1/sqrt(dot(v,v))), inversesqrt(x) = 1/sqrt(x), dot sums left to right;
 *   - function arguments evaluate left to right; `seed` parameters are by value where the GLSL says so;
 *   - pow(x,5) = ((x*x)*(x*x))*x; pow(x,y) for the sRGB curves = exp2(y*log2(x)) with the polynomial
 *     exp2/log2 below; sin/cos by Cody-Waite reduction + minimax polynomials (all in fp32);
 *   - float(uint) rounds to nearest even;
 *   - texture(): RGBA8 UNORM texel/255, REPEAT by fract(), bilinear weights in fp32 from uv*size-0.5,
 *     lerp in x then y; texture id 0 = opaque white;
 *   - image loads outside the image return 0, stores outside are dropped.
 */
#ifndef PTR_ORACLE_H_
#define PTR_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    const uint8_t* rgba; /* w*h*4 bytes, row-major, row 0 first; NULL for an unused slot */
    uint32_t w, h;
} orc_texture;

typedef struct {
    float* rgba; /* w*h*4 floats */
    uint32_t w, h;
} orc_image;

typedef struct {
    uint64_t rays;       /* hit_node invocations */
    uint64_t nodes;      /* BVH node records fetched */
    uint64_t triangles;  /* triangle tests */
    uint64_t shaded;     /* hits that reached the shading block */
    uint64_t nan_hits;   /* triangle tests accepted with a NaN t (degenerate-triangle tie class) */
} orc_counters;

/* Everything the 64-byte constant block + bindless tables give the shader. */
typedef struct {
    const void* scene_meta;      /* 160 bytes: camera(128) + 8 x u32 */
    const void* bvh_nodes;       /* packed_bvh_node[] (32 B each) */
    const uint32_t* indices;     /* 3 per triangle, index | material byte packing */
    const float* positions;      /* 3 per vertex */
    const float* normals;        /* 3 per vertex */
    const float* uvs;            /* 2 per vertex */
    const void* materials;       /* 48 B each */
    const orc_texture* textures; /* bindless sampled-image table, slot 0 unused (= white) */
    uint32_t texture_count;
    orc_image accumulation;      /* read */
    orc_image output;            /* written */
    /* optional per-pixel capture of the first two closest hits (NULL to skip): 4 values per pixel
       of the width x height dispatch: prim(bounce0), t(bounce0), prim(bounce1), t(bounce1);
       prim = -1 where the ray missed or the path had ended, -2 = NaN-t tie class */
    int32_t* aov_prim;           /* 2 per pixel */
    float* aov_t;                /* 2 per pixel */
    const void* flat_bvh;        /* optional flattened-layout blob (header|nodes|triangles) */
    int32_t use_flat;            /* !=0: closest hits come from the flattened layout's traversal */
    uint32_t index_words;        /* 0 or 3: the reference's packing (3 words per triangle, 24-bit vertex id | one byte of
                                    the material id, src/scene.cpp:140-142); 4: extension for scenes past 2^24 vertices,
                                    4 plain words per triangle: v0, v1, v2, material */
    uint32_t seed_sample_index;  /* 0: the per-pixel seed uses the metadata's sample_index (compute.comp:270); otherwise this value
                                    seeds the pixel while the metadata's sample_index still weights the accumulation: the sample
                                    partition over several GPUs, where a GPU's k-th frame draws global sample (k-1)*world+rank+1 */
    const void* flat_instances;  /* optional top level over placed copies of flat_bvh (header|nodes|instance records); used with
                                    use_flat.  Captured primitive ids then read instance * triangle_count + primitive. */
    const void* spheres;         /* optional: the reference's `sphere` records (compute.comp:31-36: vec4 position, 48-byte material,
                                    radius, 12 bytes of padding = 80 bytes) that the sphere leaves of flat_bvh name; used with use_flat */
    uint32_t sphere_count;
} orc_frame;

/* compute.comp main() over the rows [y0, y1) of the dispatch, `threads` worker threads. */
void orc_render(const orc_frame* f, uint32_t y0, uint32_t y1, int threads, orc_counters* counters);
/* tonemapping.comp main() over a gx*8 x gy*8 dispatch. */
void orc_tonemap(const orc_frame* f, uint32_t groups_x, uint32_t groups_y);

/* ---- unit entry points for known-answer tests ---- */
uint32_t orc_pcg_hash(uint32_t* seed);
float orc_rand(uint32_t* seed);
float orc_sin(float x);
float orc_cos(float x);
float orc_pow(float x, float y);
void orc_srgb_to_linear(const float in[3], float out[3]);
void orc_linear_to_srgb(const float in[3], float out[3]);
int orc_hit_aabb(const float bmin[3], const float bmax[3], const float origin[3], const float dir[3], float min_t, float max_t);
/* out: t, bary[3], geometry_normal[3] */
int orc_hit_triangle_pos(const float p0[3], const float p1[3], const float p2[3], const float origin[3],
                         const float dir[3], float min_t, float max_t, float* t, float bary[3], float gn[3]);
int orc_hit_sphere(const float center[3], float radius, const float origin[3], const float dir[3], float min_t,
                   float max_t, float* t);
void orc_rotation_to_z_axis(const float axis[3], float q[4]);
void orc_rotate_point(const float q[4], const float v[3], float out[3]);
void orc_sample_hemisphere(float r0, float r1, float out[3]);
void orc_disk_vec(float r0, float r1, float out[2]);
void orc_sample_ggx_vndf(const float ve[3], float ax, float ay, uint32_t seed, float out[3]);
float orc_brdf_probability(const float color[3], float metalness, const float view[3], const float n[3]);
void orc_sample_texture(const orc_texture* t, float u, float v, float out[4]);
void orc_camera_ray(const void* scene_meta, int32_t cx, int32_t cy, uint32_t seed, float origin[3], float dir[3]);
/* closest hit over the threaded tree: returns 1 on hit; prim/t/bary out */
int orc_closest_hit(const orc_frame* f, const float origin[3], const float dir[3], float min_t, float max_t,
                    int32_t* prim, float* t, float bary[3], orc_counters* c);
/* closest hit over the flattened layout (same conventions) */
int orc_closest_hit_flat(const void* flat_bvh, const float origin[3], const float dir[3], float min_t, float max_t,
                         int32_t* prim, float* t, float bary[3], orc_counters* c);
/* closest hit over the instances in `flat_instances` of the tree `flat_bvh` (flat_instances NULL: the tree alone) */
int orc_closest_hit_two_level(const void* flat_bvh, const void* flat_instances, const float origin[3], const float dir[3], float min_t,
                              float max_t, int32_t* prim, int32_t* inst, float* t, float bary[3], orc_counters* c);
/* hit_aabbs(): number of inner boxes hit (debug heat-map) */
uint32_t orc_hit_aabbs(const orc_frame* f, const float origin[3], const float dir[3], float min_t, float max_t);

#ifdef __cplusplus
}
#endif
#endif /* PTR_ORACLE_H_ */
