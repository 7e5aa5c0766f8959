// Effective shader clock of gfx950 under a chip-wide VALU load, and issue cost of a few VALU
// instruction kinds at 8 waves/SIMD.  clock64() = s_memtime (shader clock), wall_clock64() =
// s_memrealtime (constant 100 MHz).   hipcc --offload-arch=gfx950 -O3 clock_rate.hip -o clock_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int KIND>
__global__ __launch_bounds__(512) void k_load(float* out, long long* clk, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (KIND == 0) { x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                             x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b); }
            if (KIND == 1) { x0 = __builtin_fminf(x0, a + u); x1 = __builtin_fminf(x1, b + u); x2 = __builtin_fminf(x2, a - u); x3 = __builtin_fminf(x3, b - u);
                             x4 = __builtin_fminf(x4, a * u); x5 = __builtin_fminf(x5, b * u); x6 = __builtin_fminf(x6, a + 2 * u); x7 = __builtin_fminf(x7, b + 2 * u);
                             asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)); }
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int KIND>
int run(const char* name, int blocks, int iters) {
    float* out; long long* clk;
    CHECK(hipMalloc(&out, sizeof(float) * blocks * 512));
    CHECK(hipMalloc(&clk, sizeof(long long) * 2 * blocks));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_load<KIND>, dim3(blocks), dim3(512), 0, 0, out, clk, 16, 1.0001f, 0.5f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_load<KIND>, dim3(blocks), dim3(512), 0, 0, out, clk, iters, 1.0001f, 0.5f);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(2 * blocks);
    CHECK(hipMemcpy(h.data(), clk, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost));
    double sc = 0, sw = 0;
    for (int i = 0; i < blocks; i++) { sc += h[2 * i]; sw += h[2 * i + 1]; }
    const double mhz = sc / sw * 100.0;
    // per SIMD: blocks*8 waves over 1024 SIMDs, each wave iters*64 instructions
    const double inst_per_simd = (double)blocks * 8 / 1024.0 * iters * 64.0;
    printf("%-10s kernel %.3f ms  shader clock %.0f MHz (s_memtime/s_memrealtime)  %.2f cycles/instr/SIMD (event time x clock)\n",
           name, ms, mhz, ms * 1e-3 * mhz * 1e6 / inst_per_simd);
    hipFree(out); hipFree(clk);
    return 0;
}

int main() {
    const int blocks = 256 * 4 * 4; // 4 resident blocks of 8 waves per CU = 8 waves/SIMD, 4 rounds
    if (run<0>("v_fma_f32", blocks, 20000)) return 1;
    if (run<1>("v_min_f32", blocks, 20000)) return 1;
    return 0;
}
