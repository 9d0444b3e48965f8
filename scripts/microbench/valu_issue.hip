// Issue cost of single gfx950 vector instructions as a function of the waves resident on a SIMD: what should the traversal
// kernel's instruction counts be priced at?  Every wave runs a long unrolled run of INDEPENDENT instructions of one kind (8 chains);
// the host times the launch with HIP events and reports SIMD cycles per wave-instruction at the clock the run held
// (s_memtime ticks per s_memrealtime tick x 100 MHz, sampled inside the kernel).
//   hipcc --offload-arch=gfx950 -O3 valu_issue.hip -o valu_issue && ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kUnroll = 64;   // instructions per loop body, 8 independent chains
typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void run(float* out, unsigned long long* clocks, int iters, float b, float c, unsigned long long mask) {
    __shared__ float lds[256 * 2];
    float a[8];
    unsigned long long m[8] = {};
    v2f p[8];
    const v2f bv = { b, b }, cv = { c, c };
    const unsigned lds_addr = threadIdx.x * 4u;
    lds[threadIdx.x] = b;
    for (int j = 0; j < 8; j++) { a[j] = (float)(threadIdx.x + j); p[j] = v2f{ a[j], 1.0f }; }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < kUnroll / 8; k++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
            if (KIND == 0) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)); }
            if (KIND == 1) { asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(a[j]) : "v"(b)); }
            if (KIND == 2) { asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(a[j]) : "v"(b)); }
            if (KIND == 3) { asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)); }
            if (KIND == 4) { asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(a[j]) : "v"(b)); }
            if (KIND == 5) { asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)); }
            if (KIND == 6) { asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)); }
            if (KIND == 7) { asm volatile("v_cvt_f32_ubyte0_e32 %0, %0" : "+v"(a[j])); }
            if (KIND == 8) { asm volatile("v_cvt_f32_ubyte1_e32 %0, %0" : "+v"(a[j])); }
            if (KIND == 9) { asm volatile("v_cvt_f32_u32_e32 %0, %0" : "+v"(a[j])); }
            if (KIND == 10) { asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "s"(mask)); }
            if (KIND == 11) { asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m[j]) : "v"(a[j]), "v"(b)); }
            if (KIND == 12) { asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(a[j]) : "v"(b)); }
            if (KIND == 13) { asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[j]) : "v"(b)); }
            if (KIND == 14) { asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a[j])); }
            if (KIND == 15) { asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)); }
            if (KIND == 16) { asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)); }
            if (KIND == 17) { asm volatile("v_xor_b32_e32 %0, %0, %1" : "+v"(a[j]) : "v"(b)); }
            if (KIND == 18) { asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)); }
            if (KIND == 19) { asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(a[j]) : "v"(b), "v"(c)); }
            if (KIND == 22) { asm volatile("v_mov_b32_e32 %0, %1" : "=v"(a[j]) : "v"(b)); }
            if (KIND == 23) { asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(a[j])); }
            if (KIND == 24) { if (j & 1) asm volatile("v_cvt_f32_ubyte1_e32 %0, %0" : "+v"(a[j])); else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)); }
            if (KIND == 25) { asm volatile("ds_write_b32 %0, %1" :: "v"(lds_addr), "v"(a[j]) : "memory"); }
            if (KIND == 26) { asm volatile("ds_read_b32 %0, %1" : "=v"(a[j]) : "v"(lds_addr) : "memory"); }
            if (KIND == 27) { asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(a[j]) : "v"(b)); }
            if (KIND == 28) { asm volatile("v_min_f32_e32 %0, %0, %1" : "+v"(a[j]) : "v"(b)); }
            if (KIND == 29) { asm volatile("v_and_b32_e32 %0, %0, %1" : "+v"(a[j]) : "v"(b)); }
            if (KIND == 30) { asm volatile("v_or_b32_e32 %0, %0, %1" : "+v"(a[j]) : "v"(b)); }
            if (KIND == 31) { asm volatile("v_lshlrev_b32_e32 %0, 3, %0" : "+v"(a[j])); }
            if (KIND == 32) { asm volatile("v_lshrrev_b32_e32 %0, 3, %0" : "+v"(a[j])); }
            if (KIND == 33) { asm volatile("v_sub_u32_e32 %0, %0, %1" : "+v"(a[j]) : "v"(b)); }
            if (KIND == 34) { asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(b)); }
            if (KIND == 35) { asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1" :: "v"(a[j]), "v"(b)); }
            if (KIND == 36) { asm volatile("v_mul_u32_u24_e32 %0, %0, %1" : "+v"(a[j]) : "v"(b)); }
            if (KIND == 37) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "s"(b), "v"(c)); }
            if (KIND == 38) { if (j & 1) asm volatile("v_cvt_f32_ubyte1_e32 %0, %0" : "+v"(a[j])); else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j]) : "v"(bv), "v"(cv)); }
            if (KIND == 39) { if (j & 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)); else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j]) : "v"(bv), "v"(cv)); }
            if (KIND == 40) { if (j & 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)); else asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(a[j]) : "v"(b)); }
            if (KIND == 41) { if (j & 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)); else asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "s"(mask)); }
            if (KIND == 42) { if (j & 1) asm volatile("v_cvt_f32_ubyte1_e32 %0, %0" : "+v"(a[j])); else asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(a[j]) : "v"(b)); }
            if (KIND == 43) { if (j % 3 == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)); else asm volatile("v_cvt_f32_ubyte1_e32 %0, %0" : "+v"(a[j])); }
            if (KIND == 44) { if (j % 3 != 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)); else asm volatile("v_cvt_f32_ubyte1_e32 %0, %0" : "+v"(a[j])); }
            if (KIND == 45) { asm volatile("s_and_b64 %0, %0, %1" : "+s"(m[j]) : "s"(mask)); }
            if (KIND == 46) { if (j & 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)); else asm volatile("s_and_b64 %0, %0, %1" : "+s"(m[j]) : "s"(mask)); }
            if (KIND == 47) { asm volatile("v_ashrrev_i32_e32 %0, 31, %0" : "+v"(a[j])); }
            if (KIND == 48) { asm volatile("v_ashrrev_i32_e32 %0, %1, %0" : "+v"(a[j]) : "v"(c)); }
            if (KIND == 49) { asm volatile("v_lshlrev_b32_e32 %0, %1, %0" : "+v"(a[j]) : "v"(c)); }
            if (KIND == 50) { asm volatile("v_lshlrev_b32_e32 %0, 8, %1" : "=v"(a[j]) : "v"(b)); }
            if (KIND == 51) { asm volatile("v_not_b32_e32 %0, %0" : "+v"(a[j])); }
            if (KIND == 52) { asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)); }
            if (KIND == 53) { asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(a[j]) : "v"(b)); }
            if (KIND == 54) { asm volatile("v_mul_f32_e32 %0, 0x3f800001, %0" : "+v"(a[j])); }
            if (KIND == 55) { asm volatile("v_and_b32_e32 %0, 0xff00, %0" : "+v"(a[j])); }
            if (KIND == 56) { asm volatile("v_or_b32_e32 %0, 0x4b000000, %0" : "+v"(a[j])); }
            if (KIND == 57) { asm volatile("v_and_b32_e32 %0, 63, %0" : "+v"(a[j])); }
            if (KIND == 58) { asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(a[j]) : "s"(b)); }
            if (KIND == 59) { asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[j]) : "s"(b)); }
            if (KIND == 60) { asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(a[j]) : "s"(b)); }
            if (KIND == 61) { asm volatile("v_fma_f32 %0, -%0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)); }
            if (KIND == 62) { asm volatile("v_sub_f32_e64 %0, |%0|, %1" : "+v"(a[j]) : "v"(b)); }
            if (KIND == 63) { asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(a[j]) : "v"(b)); }
            if (KIND == 64) { asm volatile("v_subrev_u32_e32 %0, %0, %1" : "+v"(a[j]) : "v"(b)); }
            if (KIND == 65) { asm volatile("v_min_u32_e32 %0, %0, %1" : "+v"(a[j]) : "v"(b)); }
            if (KIND == 66) { asm volatile("v_cvt_f32_i32_e32 %0, %0" : "+v"(a[j])); }
            if (KIND == 67) { asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)); }
            if (KIND == 68) { asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(a[j]) : "v"(b)); }
            if (KIND == 69) { asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)); }
            if (KIND == 70) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[j]) : "v"(b)); }
            if (KIND == 71) { asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(a[j]) : "v"(b)); }
            if (KIND == 72) { asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "+v"(a[j])); }
            if (KIND == 73) { if (j % 4 == 3) asm volatile("v_cvt_f32_ubyte1_e32 %0, %0" : "+v"(a[j])); else asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(a[j]) : "v"(b)); }
            if (KIND == 74) { if (j == 0) asm volatile("ds_write_b32 %0, %1" :: "v"(lds_addr), "v"(a[j]) : "memory"); else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)); }
                if (KIND == 20) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[j]) : "v"(bv));
                if (KIND == 21) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j]) : "v"(bv), "v"(cv));
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int j = 0; j < 8; j++) s += a[j] + p[j].x + p[j].y + (float)m[j];
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = t1 - t0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
}

typedef void (*kernel_t)(float*, unsigned long long*, int, float, float, unsigned long long);
template <int K> struct Table { static void fill(kernel_t* t) { t[K] = run<K>; Table<K - 1>::fill(t); } };
template <> struct Table<-1> { static void fill(kernel_t*) {} };

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float* out; unsigned long long* clocks;
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMalloc(&clocks, 16 * 8 * cus * sizeof(unsigned long long)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char* names[] = { "v_fma_f32", "v_mul_f32", "v_sub_f32", "v_fmac_f32", "v_max_f32", "v_max3_f32", "v_med3_f32", "v_cvt_f32_ubyte0", "v_cvt_f32_ubyte1", "v_cvt_f32_u32", "v_cndmask_b32(sgpr)", "v_cmp_gt_f32(sgpr)", "v_add_u32", "v_lshl_add_u32", "v_bfe_u32", "v_and_or_b32", "v_perm_b32", "v_xor_b32", "v_mad_u32_u24", "v_fma_mix_f32(f16 src0)", "v_pk_mul_f32", "v_pk_fma_f32", "v_mov_b32", "v_rcp_f32", "fma+cvt alternating", "ds_write_b32", "ds_read_b32", "v_add_f32", "v_min_f32", "v_and_b32", "v_or_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_sub_u32", "v_cndmask_b32_e32(vcc)", "v_cmp_gt_f32_e32(vcc)", "v_mul_u32_u24", "v_fma_f32(sgpr src)", "pk_fma+cvt alternating", "pk_fma+fma alternating", "max+fma alternating", "cndmask+fma alternating", "cvt+max alternating", "cvt+cvt+fma (2:1)", "cvt+fma+fma (1:2)", "s_and_b64 (scalar)", "fma + s_and alternating", "v_ashrrev_i32 31", "v_ashrrev_i32 vgpr", "v_lshlrev_b32 vgpr", "v_lshlrev_b32 8 (src!=dst)", "v_not_b32", "v_bfi_b32", "v_fma_f32 inline 1.0", "v_mul_f32 literal", "v_and_b32 literal", "v_or_b32 literal", "v_and_b32 inline 63", "v_xor_b32 sgpr", "v_mul_f32 sgpr", "v_add_u32 sgpr", "v_fma_f32 neg mod", "v_sub_f32_e64 abs", "v_mul_f32_e64 (vop3 enc)", "v_subrev_u32", "v_min_u32", "v_cvt_f32_i32", "v_add3_u32", "v_lshl_or_b32", "v_or3_b32", "v_mul_lo_u32", "v_mov_b32 dpp quad_perm", "v_cvt_f32_u32 sdwa byte1", "v_add_f32 + v_cvt 3:1", "ds_write_b32 + 7 fma" };
    constexpr int kKinds = 75;
    kernel_t table[kKinds];
    Table<kKinds - 1>::fill(table);
    printf("%s: %d CUs; SIMD cycles per wave-instruction = launch time x in-kernel clock / wave-instructions issued on one SIMD\n", prop.name, cus);
    printf("%-26s %10s %10s %10s %12s\n", "instruction", "waves/SIMD", "clock GHz", "launch ms", "cycles/instr");
    for (int kind = (argc > 1 ? atoi(argv[1]) : 0); kind < kKinds; kind++) {
        for (int wps : { 1, 2, 6 }) {
            const int blocks = cus * wps, iters = 12000 / wps;   // blocks of 256 threads = one wave per SIMD; wps blocks per CU
            hipLaunchKernelGGL(table[kind], dim3(blocks), dim3(256), 0, 0, out, clocks, 200, 1.0001f, 0.5f, 0x5555555555555555ull);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(table[kind], dim3(blocks), dim3(256), 0, 0, out, clocks, iters, 1.0001f, 0.5f, 0x5555555555555555ull);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> h(2 * blocks);
            CHECK(hipMemcpy(h.data(), clocks, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            double ticks = 0, real = 0;
            for (int k = 0; k < blocks; k++) { ticks += (double)h[2 * k]; real += (double)h[2 * k + 1]; }
            const double ghz = ticks / real * 0.1;                       // s_memrealtime ticks at 100 MHz
            const double wave_instr_per_simd = (double)iters * kUnroll * wps;
            printf("%-26s %10d %10.3f %10.3f %12.2f\n", names[kind], wps, ghz, ms, ms * 1e-3 * ghz * 1e9 / wave_instr_per_simd);
        }
    }
    return 0;
}
