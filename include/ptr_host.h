/* C entry points of the host-side scene preparation (libptr_host.so): the BVH builder with its GPU
 * flattener and the camera, i.e. the producers of the buffers the kernels read.  They wrap the C++
 * classes that keep the reference's API (`bvh`, `camera` in path-tracer_amd/host/), for callers that
 * cannot include C++ headers (the ctypes test-suite and benchmark driver).
 * All functions return 0 / a non-negative count on success and a negative value on failure;
 * ptrh_last_error() returns the message of the calling thread's last failure. */
#ifndef PTR_HOST_H_
#define PTR_HOST_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* ptrh_last_error(void);

/* out = { sizeof(packed_bvh_node), sizeof(vec3), sizeof(camera), sizeof(triangle) } = { 32, 16, 128, 64 } */
void ptrh_sizes(uint32_t out[4]);

/* bvh::bvh(triangles, packed_nodes) -- reference: src/bvh.cpp:48-86.
 * positions: n * 9 floats (three corners per triangle); nodes_out: room for 2n-1 records of 32 bytes.
 * mode 0 = reference_order (bit-identical to the reference), 1 = sah_partitioned.  Returns the node count. */
int64_t ptrh_bvh_build(const float* positions, int64_t n, int32_t mode, void* nodes_out);

/* Position of every primitive in the threaded tree's leaf visiting order (tie-break rank of the flattened layout). */
int32_t ptrh_bvh_visit_ranks(const void* nodes, int64_t node_count, int64_t primitive_count, int32_t* ranks_out);

/* bvh::flatten(): the 64-byte-node GPU layout (header | nodes | triangles) in a malloc'ed blob; free with ptrh_free.
 * ranks may be NULL (rank = primitive index); threads 0 = automatic. */
int32_t ptrh_bvh_flatten(const float* positions, int64_t n, const int32_t* ranks, uint32_t threads, void** blob_out,
                         uint64_t* size_out);
/* The same with sphere leaves: `spheres` holds sphere_count records of 4 floats (centre, radius) -- the primitive of the
 * reference's hit_sphere (shaders/include/ray.h:105-135; its call site in hit_node is commented out, ray.h:173-177).
 * Sphere s reports primitive 0x40000000 | s; n may be 0. */
int32_t ptrh_bvh_flatten_spheres(const float* positions, int64_t n, const int32_t* ranks, const float* spheres, int64_t sphere_count, uint32_t threads,
                                 void** blob_out, uint64_t* size_out);
/* scene_source::random_spheres(): the scene of the reference's commented-out scene::random_scene (src/scene.cpp:11-58) -- a ground
 * sphere, a 22 x 22 field of small ones, three large ones -- deterministic in `seed`.  Writes up to `capacity` records of 9 floats
 * (x y z radius r g b metalness roughness) and the number of spheres the scene has (call with capacity 0 to ask). */
int32_t ptrh_random_spheres(uint32_t seed, float* out9, uint32_t capacity, uint32_t* count_out);
/* bvh::flatten_instances(): the top level over `instance_count` placed copies of one flattened tree -- an extension,
 * the reference has no instancing (its scene is one triangle soup, src/scene.cpp:84-176).  object_blob is
 * ptrh_bvh_flatten()'s (or ptr_bvh_build_flat()'s) output; object_to_world holds 12 floats per instance, the three
 * rows (3x3 part, translation) of an invertible affine matrix.  Returns header | 64-byte nodes | 64-byte instance
 * records (world-to-object rows, object root) in a malloc'ed blob for ptr_bvh_bind_instances; free with ptrh_free. */
int32_t ptrh_bvh_flatten_instances(const void* object_blob, uint64_t object_bytes, const float* object_to_world, int64_t instance_count,
                                   uint32_t threads, void** blob_out, uint64_t* size_out);
/* bvh::flatten_objects(): several objects in ONE flattened layout -- triangles [object_first[k], object_first[k+1]) form
 * object k (object_first holds object_count + 1 offsets, from 0 to n), each with a tree of its own; primitive ids stay
 * global.  objects_out receives object_count entries of 8 words: root reference (int32), stack need (uint32), box min
 * (3 floats), box max (3 floats).  The header names object 0's root. */
int32_t ptrh_bvh_flatten_objects(const float* positions, int64_t n, const int32_t* ranks, const uint32_t* object_first, int64_t object_count,
                                 uint32_t threads, void** blob_out, uint64_t* size_out, void* objects_out);
/* bvh::flatten_instances() over such objects: copy i shows objects[object_of_copy[i]] through object_to_world[12 i ..]. */
int32_t ptrh_bvh_flatten_placements(const void* objects, int64_t object_count, const uint32_t* object_of_copy, const float* object_to_world,
                                    int64_t copy_count, uint32_t threads, void** blob_out, uint64_t* size_out);
void ptrh_free(void* p);

/* camera::camera + optional set_aspect_ratio / move / rotate_y -- reference: src/camera.cpp:6-57.
 * ops: n_ops x (opcode, a, b, c): 0 = set_aspect_ratio(a), 1 = move((a,b,c)), 2 = rotate_y(a).
 * out: the 128-byte camera block that opens the scene metadata. */
void ptrh_camera(const float pos[3], const float target[3], float v_fov, float aspect, float aperture, float focus_dist,
                 const float* ops, int32_t n_ops, void* out);

/* ---- glTF -> flat scene arrays: gltf::gltf(path) + the flattening loop of scene::scene -- reference:
 * src/gltf.cpp:14-195, src/mesh.cpp:5-37, src/scene.cpp:84-176.  *scene_out is an opaque flat scene; free it with
 * ptrh_scene_free.  apply_transforms 0 = node transforms ignored (what the reference does), 1 = applied, 2 = instanced
 * (gltf::flatten_instanced, an extension: every mesh once, untransformed, and one placement per node that shows it).
 * wide_indices -1 = the reference's packed index words unless the scene has more than 2^24 vertices or material records,
 * 0 = packed (fails for such a scene), 1 = always the 4-word extension (v0, v1, v2, material per triangle). */
int32_t ptrh_gltf_load(const char* path, int32_t apply_transforms, int32_t wide_indices, void** scene_out);
/* The same flat scene from a .ptrscene file (the headless driver's native container). */
int32_t ptrh_scene_load(const char* path, void** scene_out);
/* counts = { index words in total, vertices, material records, textures, index words per triangle (3 or 4) } */
int32_t ptrh_scene_counts(const void* scene, uint64_t counts[5]);
/* Instanced scenes: counts = { objects, placements }; object_first u32[objects + 1] (triangle offsets), object_of_copy
 * u32[placements], object_to_world f32[12 * placements] (rows of 3x3 part, translation).  Array pointers may be NULL. */
int32_t ptrh_scene_placements(const void* scene, uint64_t counts[2], uint32_t* object_first, uint32_t* object_of_copy, float* object_to_world);
/* Copies the arrays out (any pointer may be NULL): indices u32[counts[0]] (3 words per triangle: 24-bit vertex id | one byte
 * of the material id, src/scene.cpp:140-142; or 4 plain words per triangle, see counts[4]), positions/normals f32[3 * vertices], uvs f32[2 * vertices], materials
 * counts[2] records of 32 bytes { f32 base_color[4]; u32 albedo_texture, mr_texture (1-based, 0 = none);
 * f32 metalness, roughness }. */
int32_t ptrh_scene_arrays(const void* scene, uint32_t* indices, float* positions, float* normals, float* uvs, void* materials);
/* Texture `index` (0-based): size and, if rgba != NULL, its width*height*4 RGBA8 texels. */
int32_t ptrh_scene_texture(const void* scene, uint32_t index, uint32_t* width, uint32_t* height, uint8_t* rgba);
void ptrh_scene_free(void* scene);

/* PNG / JPEG file image -> RGBA8 (what stbi_load(path, &w, &h, &channels, 4) returns, src/gltf.cpp:103-112);
 * *rgba_out is malloc'ed, free with ptrh_free. */
int32_t ptrh_image_decode(const uint8_t* bytes, uint64_t size, int32_t* width, int32_t* height, int32_t* channels, uint8_t** rgba_out);
/* width*height RGB8 texels -> a PNG file image (malloc'ed, free with ptrh_free): the output stage's container next to the
 * reference's PPM text (src/color.cpp:6-14 gives the 8-bit values). */
int32_t ptrh_image_encode_png(const uint8_t* rgb, uint32_t width, uint32_t height, uint8_t** file_out, uint64_t* size_out);
/* Parses a JSON text and re-serialises it compactly (malloc'ed, free with ptrh_free): the reader's test hook. */
int32_t ptrh_json_dump(const char* text, uint64_t size, char** out);

#ifdef __cplusplus
}
#endif
#endif /* PTR_HOST_H_ */
