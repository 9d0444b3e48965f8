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
void ptrh_free(void* p);

/* camera::camera + optional set_aspect_ratio / move / rotate_y -- reference: src/camera.cpp:6-57.
 * ops: n_ops x (opcode, a, b, c): 0 = set_aspect_ratio(a), 1 = move((a,b,c)), 2 = rotate_y(a).
 * out: the 128-byte camera block that opens the scene metadata. */
void ptrh_camera(const float pos[3], const float target[3], float v_fov, float aspect, float aperture, float focus_dist,
                 const float* ops, int32_t n_ops, void* out);

#ifdef __cplusplus
}
#endif
#endif /* PTR_HOST_H_ */
