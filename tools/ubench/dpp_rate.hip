// Micro-benchmark: issue cost and semantics of DPP row_newbcast operands on gfx950 (the vB-side
// record of the chunk-staged DP lives in 2 VGPRs per lane: lane l of every 16-lane row holds dwords
// l and l + 16; a subtraction takes dword k of the record as a DPP operand).
// Build: hipcc --offload-arch=gfx950 -O3 -o dpp_rate dpp_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 2048
#define UNROLL 16
__global__ __launch_bounds__(512) void k_sub_dpp(float* out, int n) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = blockIdx.x + (threadIdx.x & 15);
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
            asm volatile("v_subrev_f32_dpp %0, %4, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                         "v_subrev_f32_dpp %1, %4, %1 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
                         "v_subrev_f32_dpp %2, %4, %2 row_newbcast:11 row_mask:0xf bank_mask:0xf\n"
                         "v_subrev_f32_dpp %3, %4, %3 row_newbcast:15 row_mask:0xf bank_mask:0xf"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
__global__ __launch_bounds__(512) void k_sub_plain(float* out, int n) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = blockIdx.x + (threadIdx.x & 15);
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
            asm volatile("v_subrev_f32 %0, %4, %0\n v_subrev_f32 %1, %4, %1\n v_subrev_f32 %2, %4, %2\n v_subrev_f32 %3, %4, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
__global__ __launch_bounds__(512) void k_readlane(float* out, int n) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    int s0, s1, s2, s3;
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
            asm volatile("v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %5, 7\n v_readlane_b32 %2, %6, 11\n v_readlane_b32 %3, %7, 15"
                         : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (float)(s0 + s1 + s2 + s3);
}
__global__ void k_semantics(float* out) {
    const float v = (float)threadIdx.x;  // lane id
    float r;
    asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    out[threadIdx.x] = r;
}
template <class K> void run(const char* name, K k, float* d, double ghz) {
    const int blocks = 256 * 4, threads = 512;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, ITERS);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)ITERS * UNROLL * 4 * 8;
    printf("%-22s %8.3f ms -> %.2f cycles per wave-instruction per SIMD (at %.2f GHz)\n", name, ms,
           ms * 1e-3 * ghz * 1e9 / instr_per_simd, ghz);
}
int main() {
    float* d; hipMalloc(&d, 256 * 4 * 512 * sizeof(float));
    run("v_subrev_f32 (vgpr)", k_sub_plain, d, 2.4);
    run("v_subrev_f32_dpp bcast", k_sub_dpp, d, 2.4);
    run("v_readlane_b32", k_readlane, d, 2.4);
    hipLaunchKernelGGL(k_semantics, dim3(1), dim3(64), 0, 0, d);
    float h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("row_newbcast:5 of lane ids:");
    for (int i = 0; i < 64; i += 8) printf(" [%d]=%g", i, h[i]);
    printf("  (expected 5, 5, 21, 21, 37, 37, 53, 53)\n");
    return 0;
}
