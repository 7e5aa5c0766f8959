/* is_k_pack.hip -- compaction of the fixed-stride Section output for the final multi-GPU gather
 * ("compacted sections + per-column counts").  The reference's output buffer holds
 * max_sections = 200 slots of 32 bytes per stixel column of which 10-40 are used
 * (/root/reference/InstanceStixels/src/Stixels.cu:629-633 copies all of them to the host); over
 * xGMI only the used ones travel: per-column counts + the sections in (image, column, section)
 * order.  Three small launches, no host round trip:
 *   k_count_sections    wave per column: index of the terminator (type == -1)
 *   k_scan_counts       one workgroup: exclusive prefix over the columns, offsets[n] = total
 *   k_scatter_sections  wave per column: 32-byte sections to their packed place
 * and the inverse (k_unpack_sections) for the receiving rank. */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "instance_stixels_core.h"

#define PK_SCAN_THREADS 1024

/* wave-uniform number of sections in front of the column's terminator (<= S - 1) */
__device__ __forceinline__ int column_count(const is_section* __restrict__ col, int S, int lane) {
    for (int i0 = 0; i0 < S; i0 += 64) {
        const int i = i0 + lane;
        const bool term = i >= S || col[i].type == -1;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(term);
        if (m) return min(i0 + (int)__builtin_ctzll(m), S - 1);
    }
    return S - 1; /* no terminator: the reference asserts i < max_sections (StixelsKernels.cu:950) */
}

__global__ __launch_bounds__(256) void k_count_sections(const is_section* __restrict__ sections,
                                                        int n_columns, int S, int32_t* __restrict__ counts) {
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (col >= n_columns) return;
    const int n = column_count(sections + (size_t)col * S, S, lane);
    if (lane == 0) counts[col] = n;
}

__global__ __launch_bounds__(PK_SCAN_THREADS) void k_scan_counts(const int32_t* __restrict__ counts,
                                                                 int n_columns, int32_t* __restrict__ offsets) {
    __shared__ int s_part[PK_SCAN_THREADS];
    const int tid = threadIdx.x;
    const int per = (n_columns + PK_SCAN_THREADS - 1) / PK_SCAN_THREADS;
    const int lo = min(tid * per, n_columns), hi = min(lo + per, n_columns);
    int sum = 0;
    for (int c = lo; c < hi; c++) sum += counts[c];
    s_part[tid] = sum;
    __syncthreads();
    for (int d = 1; d < PK_SCAN_THREADS; d <<= 1) { /* Hillis-Steele inclusive scan of the partials */
        const int v = tid >= d ? s_part[tid - d] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int run = s_part[tid] - sum;
    for (int c = lo; c < hi; c++) {
        offsets[c] = run;
        run += counts[c];
    }
    if (tid == PK_SCAN_THREADS - 1) offsets[n_columns] = s_part[tid];
}

__global__ __launch_bounds__(256) void k_scatter_sections(const is_section* __restrict__ sections,
                                                          int n_columns, int S,
                                                          const int32_t* __restrict__ counts,
                                                          const int32_t* __restrict__ offsets,
                                                          is_section* __restrict__ packed) {
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (col >= n_columns) return;
    const int n = counts[col];
    const int4* src = reinterpret_cast<const int4*>(sections + (size_t)col * S);
    int4* dst = reinterpret_cast<int4*>(packed + offsets[col]);
    for (int i = lane; i < 2 * n; i += 64) dst[i] = src[i]; /* 32-byte sections as 16-byte halves */
}

__global__ __launch_bounds__(256) void k_unpack_sections(const int32_t* __restrict__ counts,
                                                         const int32_t* __restrict__ offsets,
                                                         const is_section* __restrict__ packed,
                                                         int n_columns, int S,
                                                         is_section* __restrict__ sections) {
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (col >= n_columns) return;
    const int n = min(max(counts[col], 0), S - 1);
    const int4* src = reinterpret_cast<const int4*>(packed + offsets[col]);
    int4* dst = reinterpret_cast<int4*>(sections + (size_t)col * S);
    for (int i = lane; i < 2 * n; i += 64) dst[i] = src[i];
    if (lane < 2) { /* terminator, StixelsKernels.cu:952-954 */
        const int4 t = lane == 0 ? make_int4(-1, 0, 0, 0) : make_int4(0, 0, 0, 0);
        dst[2 * n + lane] = t;
    }
}

extern "C" {

hipError_t isk_launch_pack(const is_section* sections, int n_columns, int S, int32_t* counts,
                           int32_t* offsets, is_section* packed, hipStream_t stream) {
    const dim3 grid((n_columns + 3) / 4);
    hipLaunchKernelGGL(k_count_sections, grid, dim3(256), 0, stream, sections, n_columns, S, counts);
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(PK_SCAN_THREADS), 0, stream, counts, n_columns, offsets);
    hipLaunchKernelGGL(k_scatter_sections, grid, dim3(256), 0, stream, sections, n_columns, S, counts,
                       offsets, packed);
    return hipGetLastError();
}

hipError_t isk_launch_unpack(const int32_t* counts, int32_t* offsets, const is_section* packed,
                             int n_columns, int S, is_section* sections, hipStream_t stream) {
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(PK_SCAN_THREADS), 0, stream, counts, n_columns, offsets);
    hipLaunchKernelGGL(k_unpack_sections, dim3((n_columns + 3) / 4), dim3(256), 0, stream, counts, offsets,
                       packed, n_columns, S, sections);
    return hipGetLastError();
}

} /* extern "C" */
