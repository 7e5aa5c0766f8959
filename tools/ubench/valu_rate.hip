// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU ops the DP uses.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define ITERS 2048
#define UNROLL 16
#define OP_KERNEL(NAME, DECL, ASMLINE)                                                   \
__global__ __launch_bounds__(512) void NAME(float* out, int n) {                          \
    DECL                                                                                  \
    for (int i = 0; i < n; i++) {                                                         \
        _Pragma("unroll") for (int u = 0; u < UNROLL; u++) { ASMLINE }                    \
    }                                                                                     \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)a0 + (float)a1 + (float)a2 + (float)a3; \
}
#define DECL_I int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = blockIdx.x, c = 7;
#define DECL_F float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = blockIdx.x, c = 7;
OP_KERNEL(k_sub_u32, DECL_I,
  asm volatile("v_sub_u32 %0, %0, %4\n v_sub_u32 %1, %1, %4\n v_sub_u32 %2, %2, %4\n v_sub_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
OP_KERNEL(k_min3_i32, DECL_I,
  asm volatile("v_min3_i32 %0, %0, %4, %5\n v_min3_i32 %1, %1, %4, %5\n v_min3_i32 %2, %2, %4, %5\n v_min3_i32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
OP_KERNEL(k_add_f32, DECL_F,
  asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
OP_KERNEL(k_fma_f32, DECL_F,
  asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
OP_KERNEL(k_min3_f32, DECL_F,
  asm volatile("v_min3_f32 %0, %0, %4, %5\n v_min3_f32 %1, %1, %4, %5\n v_min3_f32 %2, %2, %4, %5\n v_min3_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
OP_KERNEL(k_cvt_f32_i32, DECL_F,
  asm volatile("v_cvt_f32_i32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
OP_KERNEL(k_cndmask, DECL_F,
  asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");)
OP_KERNEL(k_cmp_lt_f32, DECL_F,
  asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cmp_lt_f32 vcc, %1, %4\n v_cmp_lt_f32 vcc, %2, %4\n v_cmp_lt_f32 vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");)

OP_KERNEL(k_subrev_sgpr, DECL_I,
  asm volatile("v_subrev_u32 %0, %4, %0\n v_subrev_u32 %1, %4, %1\n v_subrev_u32 %2, %4, %2\n v_subrev_u32 %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(n));)
OP_KERNEL(k_min_i32, DECL_I,
  asm volatile("v_min_i32 %0, %0, %4\n v_min_i32 %1, %1, %4\n v_min_i32 %2, %2, %4\n v_min_i32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
OP_KERNEL(k_min_f32, DECL_F,
  asm volatile("v_min_f32 %0, %0, %4\n v_min_f32 %1, %1, %4\n v_min_f32 %2, %2, %4\n v_min_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
OP_KERNEL(k_mul_f32, DECL_F,
  asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
OP_KERNEL(k_cndmask_s, DECL_F,
  asm volatile("v_cndmask_b32 %0, %0, %4, %5\n v_cndmask_b32 %1, %1, %4, %5\n v_cndmask_b32 %2, %2, %4, %5\n v_cndmask_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "s"((unsigned long long)n * 0x10001ull));)
OP_KERNEL(k_cvt_i32_f32, DECL_F,
  asm volatile("v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
OP_KERNEL(k_lshl_add, DECL_I,
  asm volatile("v_lshl_add_u32 %0, %0, 2, %4\n v_lshl_add_u32 %1, %1, 2, %4\n v_lshl_add_u32 %2, %2, 2, %4\n v_lshl_add_u32 %3, %3, 2, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
OP_KERNEL(k_max_f32, DECL_F,
  asm volatile("v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_max_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
OP_KERNEL(k_sub_f32_sgpr, DECL_F,
  asm volatile("v_subrev_f32 %0, %4, %0\n v_subrev_f32 %1, %4, %1\n v_subrev_f32 %2, %4, %2\n v_subrev_f32 %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"((float)n));)
OP_KERNEL(k_mov, DECL_F,
  asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
__global__ __launch_bounds__(512) void k_pk_add_f32(float* out, int n) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, b = {(float)blockIdx.x, 2.f};
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
            asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a1.y + a2.x + a3.y;
}
__global__ __launch_bounds__(512) void k_add_f64(float* out, int n) {
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = blockIdx.x;
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
            asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3);
}
__global__ __launch_bounds__(512) void k_cvt_f32_f64(float* out, int n) {
    double b0 = threadIdx.x, b1 = b0 + 1, b2 = b0 + 2, b3 = b0 + 3;
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
            asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
// EXEC-masked variants: does the SIMD skip a 32-lane half (or more) whose EXEC bits are all zero?
#define MASK_KERNEL(NAME, MASK)                                                               \
__global__ __launch_bounds__(512) void NAME(float* out, int n) {                              \
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = blockIdx.x;             \
    asm volatile("s_mov_b64 exec, %0" : : "s"((unsigned long long)(MASK)));                    \
    for (int i = 0; i < n; i++) {                                                              \
        _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {                                   \
            asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b)); \
        }                                                                                      \
    }                                                                                          \
    asm volatile("s_mov_b64 exec, -1");                                                        \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;                            \
}
MASK_KERNEL(k_add_lo32, 0x00000000FFFFFFFFull)
MASK_KERNEL(k_add_hi32, 0xFFFFFFFF00000000ull)
MASK_KERNEL(k_add_lo16, 0x000000000000FFFFull)
MASK_KERNEL(k_add_one, 0x1ull)
#define MASK_KERNEL4(NAME, MASK)                                                              \
__global__ __launch_bounds__(512) void NAME(float* out, int n) {                              \
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = blockIdx.x;             \
    asm volatile("s_mov_b64 exec, %0" : : "s"((unsigned long long)(MASK)));                    \
    for (int i = 0; i < n; i++) {                                                              \
        _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {                                   \
            asm volatile("v_min_f32 %0, %0, %4\n v_min_f32 %1, %1, %4\n v_min_f32 %2, %2, %4\n v_min_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b)); \
        }                                                                                      \
    }                                                                                          \
    asm volatile("s_mov_b64 exec, -1");                                                        \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;                            \
}
MASK_KERNEL4(k_min_lo32, 0x00000000FFFFFFFFull)
MASK_KERNEL4(k_min_hi32, 0xFFFFFFFF00000000ull)
template <class K> void run(const char* name, K k, float* d, double ghz) {
    const int blocks = 256 * 4, threads = 512;  // 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, ITERS);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)ITERS * UNROLL * 4 * 8;  // 8 waves per SIMD
    printf("%-16s %8.3f ms  -> %.2f cycles per wave-instruction per SIMD (at %.2f GHz)\n", name, ms,
           ms * 1e-3 * ghz * 1e9 / instr_per_simd, ghz);
}
int main() {
    float* d; hipMalloc(&d, 256 * 4 * 512 * sizeof(float));
    const double ghz = 2.4;
    run("v_sub_u32", k_sub_u32, d, ghz); run("v_min3_i32", k_min3_i32, d, ghz);
    run("v_add_f32", k_add_f32, d, ghz); run("v_fma_f32", k_fma_f32, d, ghz);
    run("v_min3_f32", k_min3_f32, d, ghz); run("v_cvt_f32_i32", k_cvt_f32_i32, d, ghz);
    run("v_cndmask_b32", k_cndmask, d, ghz); run("v_cmp_lt_f32", k_cmp_lt_f32, d, ghz);
    run("v_pk_add_f32", k_pk_add_f32, d, ghz); run("v_add_f64", k_add_f64, d, ghz);
    run("v_cvt_f32_f64", k_cvt_f32_f64, d, ghz);
    run("v_subrev_u32 sgpr", k_subrev_sgpr, d, ghz); run("v_min_i32", k_min_i32, d, ghz);
    run("v_min_f32", k_min_f32, d, ghz); run("v_mul_f32", k_mul_f32, d, ghz);
    run("v_cndmask sgpr", k_cndmask_s, d, ghz); run("v_cvt_i32_f32", k_cvt_i32_f32, d, ghz);
    run("v_lshl_add_u32", k_lshl_add, d, ghz); run("v_max_f32", k_max_f32, d, ghz);
    run("v_subrev_f32 sgpr", k_sub_f32_sgpr, d, ghz); run("v_mov_b32", k_mov, d, ghz);

    run("v_add_f32 exec lo32", k_add_lo32, d, ghz); run("v_add_f32 exec hi32", k_add_hi32, d, ghz);
    run("v_add_f32 exec lo16", k_add_lo16, d, ghz); run("v_add_f32 exec 1 lane", k_add_one, d, ghz);
    run("v_min_f32 exec lo32", k_min_lo32, d, ghz); run("v_min_f32 exec hi32", k_min_hi32, d, ghz);
    return 0;
}
