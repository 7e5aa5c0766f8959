#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
template <int K> __device__ __forceinline__ float dpp_sub(float mine, float R) {
    float d;
    asm volatile("v_subrev_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(R), "v"(mine), "n"(K));
    return d;
}
template <int K> __device__ __forceinline__ int dpp_sub_i(int mine, float R) {
    int d;
    asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(R), "n"(K));
    return mine - d;
}
__global__ void k(const float* rec, float* out, int* outi) {
    const int lane = threadIdx.x;
    const float R = rec[lane & 15];
    const float mine = 1000.0f + lane;
    out[lane] = dpp_sub<5>(mine, R);
    out[64 + lane] = dpp_sub<15>(mine, R);
    outi[lane] = dpp_sub_i<3>(100000 + lane, R);
}
int main() {
    float h[16]; for (int i = 0; i < 16; i++) h[i] = 10.0f * i;
    int hi[16]; for (int i = 0; i < 16; i++) hi[i] = 7 * i; 
    float* d; float* o; int* oi;
    hipMalloc(&d, 64); hipMalloc(&o, 512); hipMalloc(&oi, 256);
    // dword 3 as int 21
    memcpy(&h[3], &hi[3], 4);
    hipMemcpy(d, h, 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, oi);
    float r[128]; int ri[64];
    hipMemcpy(r, o, 512, hipMemcpyDeviceToHost); hipMemcpy(ri, oi, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) {
        if (r[l] != 1000.0f + l - 50.0f) bad++;
        if (r[64 + l] != 1000.0f + l - 150.0f) bad++;
        if (ri[l] != 100000 + l - 21) bad++;
    }
    printf("bad = %d  (r[0]=%g r[17]=%g r[64]=%g ri[5]=%d)\n", bad, r[0], r[17], r[64], ri[5]);
    return bad != 0;
}
