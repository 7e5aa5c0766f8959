/* is_k_frontend.hip -- kernels around the DP: JoinColumns, the CNN output wrapper, the
 * v-disparity histogram of the road estimation.  See is_kernels.h. */
#include "is_kernels.h"

/* ====================================================================================== */
/* A3  JoinColumns                                                                         */
/* ====================================================================================== */
#define JOIN_ROWS 64
#define JOIN_COLS 32

__device__ __forceinline__ float join_median(float* tmp_row, int n) {
    /* partial selection sort exactly as StixelsKernels.cu:1007-1022 / 1038-1053 */
    for (int i = 0; i < (n / 2) + 1; i++) {
        int min_idx = i;
        for (int j = i + 1; j < n; j++)
            if (tmp_row[j] < tmp_row[min_idx]) min_idx = j;
        const float tmp = tmp_row[i];
        tmp_row[i] = tmp_row[min_idx];
        tmp_row[min_idx] = tmp;
    }
    float median = tmp_row[n / 2];
    if (n % 2 == 0) median = (median + tmp_row[(n / 2) - 1]) / 2.0f;
    return median;
}

__device__ __forceinline__ float join_one(const float* __restrict__ src, int step, bool median,
                                          float invalid) {
    float v[16];
    if (step == 8 && (((uintptr_t)src) & 15) == 0) {
        /* the usual stixel width: the eight values as two 16-byte loads instead of eight dword loads */
        const float4 a = reinterpret_cast<const float4*>(src)[0], b = reinterpret_cast<const float4*>(src)[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
#pragma unroll
        for (int i = 8; i < 16; i++) v[i] = 0.0f;
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = (i < step) ? src[i] : 0.0f;
    }
    if (median) {
        if (invalid >= 0) {
            float t[16];
            int n = 0;
#pragma unroll
            for (int i = 0; i < 16; i++)
                if (i < step && v[i] != invalid) t[n++] = v[i];
            return (n > 0) ? join_median(t, n) : invalid;
        }
        return join_median(v, step);
    }
    float mean = 0.0f;
    if (invalid >= 0) {
        int bad = 0;
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (i < step) {
                if (v[i] != invalid) mean += v[i]; else bad++;
            }
        return (bad != step) ? mean / (float)(step - bad) : invalid;
    }
#pragma unroll
    for (int i = 0; i < 16; i++)
        if (i < step) mean += v[i];
    return mean / (float)step;
}

__global__ __launch_bounds__(256) void k_join_columns(const float* __restrict__ big,
                                                      float* __restrict__ joined, int H, int W,
                                                      int C, int step, int margin, int median,
                                                      float invalid) {
    __shared__ float tile[JOIN_COLS][JOIN_ROWS + 1];
    const int img = blockIdx.z;
    const int row0 = blockIdx.x * JOIN_ROWS;
    const int col0 = blockIdx.y * JOIN_COLS;
    const float* src = big + (size_t)img * H * W;
    float* dst = joined + (size_t)img * C * H;
    const int tx = threadIdx.x % JOIN_COLS, ty = threadIdx.x / JOIN_COLS; /* 32 x 8 */
    for (int r = ty; r < JOIN_ROWS; r += 256 / JOIN_COLS) {
        const int row = row0 + r, col = col0 + tx;
        float val = 0.0f;
        if (row < H && col < C)
            val = join_one(src + (size_t)row * W + col * step + margin, step, median != 0, invalid);
        tile[tx][r] = val;
    }
    __syncthreads();
    const int rr = threadIdx.x % JOIN_ROWS, cc = threadIdx.x / JOIN_ROWS; /* 64 x 4 */
    for (int c = cc; c < JOIN_COLS; c += 256 / JOIN_ROWS) {
        const int row = row0 + rr, col = col0 + c;
        if (row < H && col < C) dst[(size_t)col * H + (H - 1 - row)] = tile[c][rr];
    }
}

/* ====================================================================================== */
/* f4  CNN output -> DP input layout ("FlipAndPad", tools/CNN_training/models/wrappers.py:35-61) */
/* ====================================================================================== */
/* in  [n][CH][Hs][Ws] float (NCHW network output: 19 x -log-softmax, 2 offset channels)
 * out [n][Ws][CH][P2S] int32: permute(0,3,1,2), rows flipped (index 0 = image bottom), zero
 * padded to P2S, value = (int)(8 * x) (truncation toward zero, as torch's .int()).
 * A 64x64 (w, k) tile per channel is transposed through LDS so that reads run along w and
 * writes along k: both sides are coalesced. */
__global__ __launch_bounds__(256) void k_flip_and_pad(const float* __restrict__ in,
                                                      int32_t* __restrict__ out, int CH, int Hs,
                                                      int Ws, int P2S) {
    __shared__ int32_t tile[64][65];
    const int n = blockIdx.z / CH, c = blockIdx.z % CH;
    const int w0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    const float* src = in + ((size_t)n * CH + c) * Hs * Ws;
    int32_t* dst = out + (size_t)n * Ws * CH * P2S + (size_t)c * P2S;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int kk = ty; kk < 64; kk += 4) { /* read: lanes along w */
        const int k = k0 + kk, w = w0 + tx;
        int32_t v = 0;
        if (k < Hs && w < Ws) v = (int32_t)(src[(size_t)(Hs - 1 - k) * Ws + w] * 8.0f);
        tile[kk][tx] = v;
    }
    __syncthreads();
    for (int ww = ty; ww < 64; ww += 4) { /* write: lanes along k */
        const int w = w0 + ww, k = k0 + tx;
        if (w < Ws && k < P2S) dst[(size_t)w * CH * P2S + k] = tile[tx][ww];
    }
}

/* ====================================================================================== */
/* f3  road estimation: v-disparity histogram, maximum, binarisation                       */
/*     (RoadEstimationKernels.cu:25-60)                                                     */
/* ====================================================================================== */
/* The reference does one global atomicAdd per pixel and a second kernel of global atomicMax.
 * Here one workgroup owns one image row: the row is read coalesced, binned with LDS atomics,
 * written once, and its maximum goes to a single global atomicMax.  Integer counts and maxima
 * do not depend on the order, so the result is identical. */
__global__ __launch_bounds__(256) void k_vdisp_histogram(const float* __restrict__ disparity,
                                                         int* __restrict__ vdisp,
                                                         int* __restrict__ maximum, int cols,
                                                         int max_dis) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* bins = (int*)smem; /* [max_dis] */
    const int row = blockIdx.x;
    for (int i = threadIdx.x; i < max_dis; i += blockDim.x) bins[i] = 0;
    __syncthreads();
    const float* src = disparity + (size_t)row * cols;
    for (int j = threadIdx.x; j < cols; j += blockDim.x) {
        const float d = src[j];
        if (d != 0) { /* RoadEstimationKernels.cu:33-37 */
            const int col = (int)d;
            if (col >= 0 && col < max_dis) atomicAdd(&bins[col], 1); /* guard: reference is unchecked */
        }
    }
    __syncthreads();
    int m = 0;
    for (int i = threadIdx.x; i < max_dis; i += blockDim.x) {
        const int v = bins[i];
        vdisp[(size_t)row * max_dis + i] = v;
        m = max(m, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(maximum, m);
}

__global__ __launch_bounds__(256) void k_vdisp_binarize(const int* __restrict__ vdisp,
                                                        uint8_t* __restrict__ out,
                                                        const int* __restrict__ maximum,
                                                        float threshold, int n) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) {
        const float p = (float)vdisp[idx]; /* RoadEstimationKernels.cu:55-58 */
        out[idx] = (p > (*maximum) * threshold) ? 255 : 0;
    }
}

extern "C" {

hipError_t isk_launch_join(const float* big, float* joined, int H, int W, int C, int step,
                           int margin, int median, float invalid, int n_images,
                           hipStream_t stream) {
    dim3 grid((H + JOIN_ROWS - 1) / JOIN_ROWS, (C + JOIN_COLS - 1) / JOIN_COLS, n_images);
    hipLaunchKernelGGL(k_join_columns, grid, dim3(256), 0, stream, big, joined, H, W, C, step,
                       margin, median, invalid);
    return hipGetLastError();
}

hipError_t isk_launch_flip_and_pad(const float* in, int32_t* out, int n, int CH, int Hs, int Ws,
                                   int P2S, hipStream_t stream) {
    dim3 grid((Ws + 63) / 64, (P2S + 63) / 64, n * CH);
    hipLaunchKernelGGL(k_flip_and_pad, grid, dim3(256), 0, stream, in, out, CH, Hs, Ws, P2S);
    return hipGetLastError();
}

hipError_t isk_launch_vdisparity(const float* disparity, int* vdisp, int* maximum, uint8_t* binary,
                                 int rows, int cols, int max_dis, float threshold,
                                 hipStream_t stream) {
    hipError_t e = hipMemsetAsync(maximum, 0, sizeof(int), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_vdisp_histogram, dim3(rows), dim3(256), sizeof(int) * max_dis, stream,
                       disparity, vdisp, maximum, cols, max_dis);
    const int n = rows * max_dis;
    hipLaunchKernelGGL(k_vdisp_binarize, dim3((n + 255) / 256), dim3(256), 0, stream, vdisp, binary,
                       maximum, threshold, n);
    return hipGetLastError();
}

} /* extern "C" */
