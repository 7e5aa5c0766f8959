// Streaming-store bandwidth of one MI355X (upper bound for k_object_lut, which writes 134 MB/frame).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_fill(float4* p, size_t n, float v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = make_float4(v, v, v, v);
}
__global__ void k_fill_rows(float* p, size_t nrows, float v) {  // one wave writes 256-byte half rows at a 512-byte stride
    const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const size_t col = w >> 1, half = w & 1;
    float* base = p + col * 1025 * 128 + half * 64 + lane;
    for (int r = 0; r < 1025; r++) base[(size_t)r * 128] = v + r;
}
int main() {
    const size_t bytes = (size_t)64 * 256 * 1025 * 128 * 4;  // lutT of 64 frames
    float4* p;
    if (hipMalloc(&p, bytes) != hipSuccess) return 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_fill, dim3(256 * 32), dim3(256), 0, 0, p, bytes / 16, 1.0f);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("float4 grid-stride fill: %.3f ms  %.2f TB/s\n", ms, bytes / ms / 1e9);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_fill_rows, dim3(64 * 256 * 2 / 4), dim3(256), 0, 0, (float*)p, (size_t)0, 2.0f);
        hipEventRecord(e1); hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
        printf("lutT-pattern fill (256 B per wave-store, 512 B stride): %.3f ms  %.2f TB/s\n", ms, bytes / ms / 1e9);
    }
    return 0;
}
