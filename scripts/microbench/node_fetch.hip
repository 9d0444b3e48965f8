// What bounds a traversal-like loop on gfx950: every lane of a wave follows its own chain through a table of 64-byte records
// (next record = a function of the bytes just read, like a BVH walk), reads Q sixteen-byte quarters of the record per step and
// issues V vector instructions on them.  Time per step against Q (4, 3, 2, 1) at fixed V tells whether the loop is bound by the
// rate of L1 accesses (one per lane and quarter: time follows Q) or by instruction issue (time follows V).
//   hipcc --offload-arch=gfx950 -O3 node_fetch.hip -o node_fetch && ./node_fetch [table MB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int Q, int V>
__global__ __launch_bounds__(256, 6) void walk(const uint4* table, uint32_t mask, int steps, uint32_t* out, int active_mod) {
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u & mask;
    float acc0 = 1.0f, acc1 = 2.0f, acc2 = 3.0f, acc3 = 4.0f;
    const bool active = (threadIdx.x % 10) < (unsigned)active_mod;   // lanes without work: 70 % active at active_mod 7
    if (active)
    for (int s = 0; s < steps; s++) {
        const uint4* rec = table + 4u * (size_t)idx;
        uint4 q[4] = {};
#pragma unroll
        for (int k = 0; k < Q; k++) q[k] = rec[k];
        uint32_t h = idx * 747796405u + 2891336453u;
#pragma unroll
        for (int k = 0; k < Q; k++) h ^= q[k].x + q[k].y + q[k].z + q[k].w;
        // V vector instructions in the mix of the node step: 2/3 of the one-pipe class (conversions, min/max, selects), 1/3 fma
        float f = __uint_as_float((h & 0x007fffffu) | 0x3f800000u);
#pragma unroll
        for (int v = 0; v < V / 6; v++) {
            acc0 = fmaf(acc0, f, acc1);
            acc1 = fmaxf(acc1, acc2 * 0.5f);
            acc2 = (float)((__float_as_uint(acc2) >> 8) & 0xffu) + acc3;
            acc3 = acc0 > acc3 ? acc1 : acc3;
            acc1 = fminf(acc1, 8.0f);
            acc0 = fmaf(acc0, 0.25f, 1.0f);
        }
        idx = (h ^ __float_as_uint(acc0)) & mask;
    }
    if (acc0 + acc1 + acc2 + acc3 == 12345.0f) out[0] = idx;
    if (idx == 0xffffffffu) out[1] = 1;
}

template <int Q, int V>
float run(const uint4* table, uint32_t mask, int steps, uint32_t* out, int blocks, int active_mod) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((walk<Q, V>), dim3(blocks), dim3(256), 0, 0, table, mask, steps / 10, out, active_mod);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((walk<Q, V>), dim3(blocks), dim3(256), 0, 0, table, mask, steps, out, active_mod);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? atoi(argv[1]) : 32;
    size_t records = 1;
    while (records * 2 * 64 <= mb << 20) records *= 2;
    std::vector<uint32_t> host(records * 16);
    uint32_t s = 12345u;
    for (auto& v : host) { s = s * 1664525u + 1013904223u; v = s; }
    uint4* table; uint32_t* out;
    CHECK(hipMalloc(&table, records * 64)); CHECK(hipMalloc(&out, 64));
    CHECK(hipMemcpy(table, host.data(), records * 64, hipMemcpyHostToDevice));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 6, steps = 2000;
    printf("table %zu MB (%zu records of 64 B), %d blocks of 256 threads (6 per CU), %d steps per lane, 7 of 10 lanes active\n", records * 64 >> 20, records, blocks, steps);
    printf("%-28s %10s %16s\n", "quarters read, instructions", "ms", "ns per wave-step");
#define ROW(Q, V) { const float ms = run<Q, V>(table, (uint32_t)records - 1u, steps, out, blocks, 7); printf("Q = %d, V = %3d               %10.3f %16.1f\n", Q, V, ms, ms * 1e6 / steps / 6.0); }
    ROW(4, 120) ROW(3, 120) ROW(2, 120) ROW(1, 120)
    ROW(4, 150) ROW(3, 150)
    ROW(4, 60) ROW(3, 60) ROW(2, 60)
    ROW(4, 0) ROW(3, 0) ROW(2, 0) ROW(1, 0)
    return 0;
}
