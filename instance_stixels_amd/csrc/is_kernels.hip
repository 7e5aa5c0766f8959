/*
 * is_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the Instance-Stixels
 * column DP.  Nothing here is translated from the reference's CUDA: the reference runs one
 * 1024-thread block per column with a barrier per vB and 126 global loads per (vB, vT) pair
 * (/root/reference/InstanceStixels/src/StixelsKernels.cu:600-839); this file restructures
 * the same arithmetic (bit-exactly, see DESIGN.md "Exact rewrites") as
 *
 *   k_join_columns      A3   StixelsKernels.cu:980-1095   LDS-transposed, coalesced both ways
 *   k_prepare_columns   A4-A6 StixelsKernels.cu:236-296, 371-469, 959-978 and
 *                            StixelsKernels.h:73-103: per-row boundary records + object LUT
 *   k_prior_tables      A9   StixelsKernels.cu:88-199 (DP-state independent part)
 *   k_dp_unary          A7-A9 StixelsKernels.cu:477-839, PAIRWISE=false: independent
 *                            (column, 64-row tile) work items, one lane per vT, vB-side
 *                            operands in SGPRs via scalar loads, vT-side LUT rows in LDS
 *   k_pw_phase1/2       A7-A9 PAIRWISE=true: per 64-row tile, a parallel launch for segments
 *                            starting in earlier tiles + a one-wave-per-column diagonal walk
 *   k_backtrace         A10  StixelsKernels.cu:843-955
 *   k_compact_instances A10  StixelsKernels.cu:926-942 in canonical order
 *
 * Numerics contract: IEEE fp32, no contraction (-ffp-contract=off), correctly rounded
 * division, no fast-math; integer sums in wrapping int32 / int64 like the reference.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "instance_stixels_core.h"
#include "is_device.h"
#include "is_numerics.h"

#define IS_INF (__builtin_inff())

typedef const __attribute__((address_space(4))) RowRec* crec_t;     /* scalar-load view */
typedef const __attribute__((address_space(4))) PriorRec* cprior_t;

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
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = (i < step) ? src[i] : 0.0f;
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
/* A4-A6  per-column preparation                                                           */
/* ====================================================================================== */
#define PREP_THREADS 256

/* Exclusive prefix of index i (0 <= i < n) with the association of the reference's
 * work-efficient block scan ComputePrefixSum (StixelsKernels.h:73-103): the up-sweep builds a
 * pairwise tree, the down-sweep gives a right child `parent + left subtree sum`, i.e. the
 * left-sibling sums on the root-to-leaf path are added top-down starting from 0.
 * pyr holds the tree: level b (n>>b nodes) at offset 2n - (2n>>b). */
__device__ __forceinline__ float blelloch_prefix(const float* pyr, int n, int log2n, int i) {
    float acc = 0.0f;
    for (int b = log2n - 1; b >= 0; b--) {
        const int node = i >> b;
        if (node & 1) acc = acc + pyr[(2 * n - ((2 * n) >> b)) + node - 1];
    }
    return acc;
}

__device__ __forceinline__ void blelloch_build(float* pyr, int n, int log2n) {
    for (int b = 1; b <= log2n; b++) {
        const float* lo = pyr + (2 * n - ((2 * n) >> (b - 1)));
        float* hi = pyr + (2 * n - ((2 * n) >> b));
        for (int j = threadIdx.x; j < (n >> b); j += PREP_THREADS) hi[j] = lo[2 * j + 1] + lo[2 * j];
        __syncthreads();
    }
}

/* Exact exclusive block scan of one int64 per thread (any association is exact). */
__device__ __forceinline__ int64_t block_excl_scan_i64(int64_t v, int64_t* s_wave /*[4]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t inc = v;
#pragma unroll
    for (int j = 1; j < 64; j <<= 1) {
        const int64_t n = __shfl_up(inc, j, 64);
        if (lane >= j) inc += n;
    }
    __syncthreads(); /* s_wave may still be read by the previous call */
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int64_t base = 0;
    for (int w = 0; w < wave; w++) base += s_wave[w];
    return base + inc - v;
}

__device__ __forceinline__ float data_cost_sky(float d, const DevParams& P) {
    /* GetDataCostSky, StixelsKernels.cu:201-215 */
    float data_cost = P.pnex_sky_log;
    if (d != P.invalid) {
        const float pgaussian = P.norm_sky + d * d * P.inv_sigma2_sky;
        const float p_data = __builtin_fminf(P.puniform_sky, pgaussian);
        data_cost = p_data + P.nopnex_sky_log;
    }
    return data_cost;
}
__device__ __forceinline__ float data_cost_ground(float fn, float d, float norm_g, float inv_s2_g,
                                                  const DevParams& P) {
    /* GetDataCostGround, StixelsKernels.cu:217-234 */
    float data_cost = P.pnex_gnd_log;
    if (d != P.invalid) {
        const float model_diff = (d - fn);
        const float pgaussian = norm_g + model_diff * model_diff * inv_s2_g;
        const float p_data = __builtin_fminf(P.puniform, pgaussian);
        data_cost = p_data + P.nopnex_gnd_log;
    }
    return data_cost;
}

/* full-resolution prefix from the 1/8-resolution exclusive prefix ps (see RowRec) */
__device__ __forceinline__ int32_t full_prefix(const int32_t* ps, int v) {
    const int k = v >> 3, m = v & 7;
    int32_t r = (int32_t)((uint32_t)ps[k] * 8u);
    if (m) r = (int32_t)((uint32_t)r + (uint32_t)(ps[k + 1] - ps[k]) * (uint32_t)m);
    return r;
}

__device__ __forceinline__ void store_instance_prefix(RowRec* o, int slow, int64_t mx, int64_t my,
                                                      int64_t mx2, int64_t my2) {
    /* dwords 24..31 of the record as two 16-byte stores */
    int4 a, b;
    if (slow) { /* RowRecWide: four int64 */
        a = make_int4((int)(uint32_t)mx, (int)(uint32_t)((uint64_t)mx >> 32), (int)(uint32_t)my,
                      (int)(uint32_t)((uint64_t)my >> 32));
        b = make_int4((int)(uint32_t)mx2, (int)(uint32_t)((uint64_t)mx2 >> 32), (int)(uint32_t)my2,
                      (int)(uint32_t)((uint64_t)my2 >> 32));
    } else { /* exact fp32 encodings, see RowRec */
        const int64_t lo_mask = ((int64_t)1 << IS_FAST_SPLIT_BITS) - 1;
        a = make_int4(__float_as_int((float)mx), __float_as_int((float)my),
                      __float_as_int((float)(mx2 - (mx2 & lo_mask))), __float_as_int((float)(mx2 & lo_mask)));
        b = make_int4(__float_as_int((float)(my2 - (my2 & lo_mask))), __float_as_int((float)(my2 & lo_mask)),
                      0, 0);
    }
    int4* d = reinterpret_cast<int4*>(o);
    d[6] = a;
    d[7] = b;
}

__global__ __launch_bounds__(PREP_THREADS) void k_prepare_columns(
    const DevParams P, const float* __restrict__ joined, const int32_t* __restrict__ seg,
    const float* __restrict__ ground /*[img][3][H]*/, const int* __restrict__ vhor_arr,
    RowRec* __restrict__ recs, int* __restrict__ col_flags, float* __restrict__ sv_arr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = P.H, P2 = P.P2, P2S = P.P2S, CH = P.CH, K = P.K;
    float* s_d = (float*)smem;                          /* [P2]   disparity column        */
    float* s_pyr = s_d + P2;                            /* [2*P2] scan tree                */
    int32_t* s_seg = (int32_t*)(s_pyr + 2 * P2);        /* [CH][P2S]                       */
    int64_t* s_wave = (int64_t*)(s_seg + CH * P2S);     /* [4]                             */

    const int colg = blockIdx.x;
    const int img = colg / P.C, col = colg % P.C;
    const int vhor = vhor_arr[img];
    const float* gfun = ground + (size_t)img * 3 * H;
    const float* gnorm = gfun + H;
    const float* gis2 = gnorm + H;
    const float* dcol = joined + (size_t)colg * H;
    const int32_t* scol = seg + (size_t)colg * CH * P2S;
    RowRec* rcol = recs + (size_t)colg * (H + 1);
    const int tid = threadIdx.x;

    for (int i = tid; i < P2; i += PREP_THREADS) s_d[i] = (i < H) ? dcol[i] : 0.0f;
    for (int i = tid; i < CH * P2S; i += PREP_THREADS) s_seg[i] = scol[i];
    __syncthreads();

    /* ---- instance-centre values per row from the RAW offsets (StixelsKernels.cu:401-409);
     * thread t owns rows [t*R, t*R+R).  mx = 8*col + 3.5 + offx + 0.5 is an exact integer;
     * my = trunc(row - offy + 0.5): n for n >= 0, n + 1 for n < 0 (truncation toward zero). */
    const int R = (H + PREP_THREADS - 1) / PREP_THREADS;
    const int r_lo = tid * R;
    const int32_t* offy = s_seg + K * P2S;
    const int32_t* offx = s_seg + (K + 1) * P2S;
    int64_t sum_mx = 0, sum_my = 0, sum_mx2 = 0, sum_my2 = 0;
    uint64_t abs_mx = 0, abs_my = 0;
    int slow = 0; /* column needs the generic (int64 / IEEE-division) DP path, see RowRec */
    for (int r = r_lo; r < r_lo + R && r < H; r++) {
        const double fx = ((double)(P.column_step * col) + 0.5 * ((double)P.column_step - 1.0)) +
                          (double)offx[r >> 3] + 0.5;
        const int64_t mx = (int64_t)fx;
        const int32_t n32 = (int32_t)((uint32_t)r - (uint32_t)offy[r >> 3]);
        const int64_t my = (int64_t)((double)n32 + 0.5);
        abs_mx += (uint64_t)(mx < 0 ? -mx : mx);
        abs_my += (uint64_t)(my < 0 ? -my : my);
        const float ad = __builtin_fabsf(s_d[r]);
        slow |= !((ad == 0.0f) || (ad >= IS_FAST_DISP_MIN && ad <= IS_FAST_DISP_MAX));
        sum_mx += mx;
        sum_my += my;
        sum_mx2 = (int64_t)((uint64_t)sum_mx2 + (uint64_t)mx * (uint64_t)mx);
        sum_my2 = (int64_t)((uint64_t)sum_my2 + (uint64_t)my * (uint64_t)my);
    }
    { /* instance centres of a FAST column: sum|mx|, sum|my| < 2^23 (block totals through LDS) */
        unsigned long long* s_abs = (unsigned long long*)s_wave;
        if (tid < 2) s_abs[tid] = 0ull;
        __syncthreads();
        atomicAdd(&s_abs[0], (unsigned long long)abs_mx);
        atomicAdd(&s_abs[1], (unsigned long long)abs_my);
        __syncthreads();
        slow |= (s_abs[0] >= (unsigned long long)IS_FAST_INSTANCE_LIMIT) |
                (s_abs[1] >= (unsigned long long)IS_FAST_INSTANCE_LIMIT);
        __syncthreads(); /* s_wave is reused by the scans below */
    }
    if (tid < K) { /* class channels of a FAST column: values >= 0, full-resolution total < 2^24 */
        const int32_t* ch = s_seg + tid * P2S;
        uint64_t total = 0;
        int negative = 0;
        for (int k = 0; k < P2S; k++) {
            negative |= (ch[k] < 0);
            total += (uint64_t)(uint32_t)ch[k];
        }
        slow |= negative | (total * IS_DOWNSAMPLE_FACTOR >= (uint64_t)IS_FAST_CLASS_LIMIT);
    }
    slow = __syncthreads_or(slow);
    if (tid == 0) col_flags[colg] = slow;
    int64_t base_mx = block_excl_scan_i64(sum_mx, s_wave);
    int64_t base_my = block_excl_scan_i64(sum_my, s_wave);
    int64_t base_mx2 = block_excl_scan_i64(sum_mx2, s_wave);
    int64_t base_my2 = block_excl_scan_i64(sum_my2, s_wave);
    /* the owner of rows [r_lo, r_lo+R) writes the exclusive prefix at those indices; the owner
     * of row H-1 also writes index H (the total) */
    for (int r = r_lo; r < r_lo + R && r < H; r++) {
        store_instance_prefix(rcol + r, slow, base_mx, base_my, base_mx2, base_my2);
        const double fx = ((double)(P.column_step * col) + 0.5 * ((double)P.column_step - 1.0)) +
                          (double)offx[r >> 3] + 0.5;
        const int64_t mx = (int64_t)fx;
        const int32_t n32 = (int32_t)((uint32_t)r - (uint32_t)offy[r >> 3]);
        const int64_t my = (int64_t)((double)n32 + 0.5);
        base_mx += mx;
        base_my += my;
        base_mx2 = (int64_t)((uint64_t)base_mx2 + (uint64_t)mx * (uint64_t)mx);
        base_my2 = (int64_t)((uint64_t)base_my2 + (uint64_t)my * (uint64_t)my);
    }
    if (r_lo <= H - 1 && H - 1 < r_lo + R)
        store_instance_prefix(rcol + H, slow, base_mx, base_my, base_mx2, base_my2);
    __syncthreads();

    /* ---- square the offset channels in place (StixelsKernels.cu:411-416), then exclusive
     * prefix of every channel at 1/8 resolution (:462-469); integer, so any order is exact */
    for (int i = tid; i < 2 * P2S; i += PREP_THREADS) {
        const uint32_t x = (uint32_t)s_seg[K * P2S + i];
        s_seg[K * P2S + i] = (int32_t)(x * x);
    }
    __syncthreads();
    /* one wave per channel (round robin): lane l owns P2S/64 consecutive entries */
    {
        const int lane = tid & 63, wv = tid >> 6, per = P2S >> 6;
        for (int c = wv; c < CH; c += PREP_THREADS / 64) {
            int32_t* ch = s_seg + c * P2S;
            if (per >= 1) {
                uint32_t local = 0;
                for (int k = 0; k < per; k++) local += (uint32_t)ch[lane * per + k];
                uint32_t inc = local; /* inclusive wave scan of the lane totals */
#pragma unroll
                for (int j = 1; j < 64; j <<= 1) {
                    const uint32_t n = (uint32_t)__shfl_up((int)inc, j, 64);
                    if (lane >= j) inc += n;
                }
                uint32_t run = inc - local;
                for (int k = 0; k < per; k++) {
                    const uint32_t x = (uint32_t)ch[lane * per + k];
                    ch[lane * per + k] = (int32_t)run;
                    run += x;
                }
            } else if (lane == 0) { /* P2S < 64: tiny columns */
                uint32_t run = 0;
                for (int k = 0; k < P2S; k++) {
                    const uint32_t x = (uint32_t)ch[k];
                    ch[k] = (int32_t)run;
                    run += x;
                }
            }
        }
    }
    __syncthreads();
    /* dwords 0..19 of every record (class prefixes + squared-offset prefix) as five 16-byte
     * chunks: consecutive threads write consecutive chunks */
    for (int i = tid; i < (H + 1) * 5; i += PREP_THREADS) {
        const int v = i / 5, q = i - v * 5;
        int32_t x[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int dw = q * 4 + j; /* dword of RowRec: Fg0 Fg1 Fon[8] Foi[8] Fsky Fnic */
            if (dw == 19) {
                x[j] = (int32_t)((uint32_t)full_prefix(s_seg + (K + 1) * P2S, v) +
                                 (uint32_t)full_prefix(s_seg + K * P2S, v));
            } else {
                const int chn = dw < 10 ? dw : (dw < 18 ? dw + 1 : 10);
                const int32_t f = full_prefix(s_seg + chn * P2S, v);
                x[j] = slow ? f : __float_as_int((float)f);
            }
        }
        reinterpret_cast<int4*>(rcol + v)[q] = make_int4(x[0], x[1], x[2], x[3]);
    }

    /* ---- fp32 prefixes with the reference's block-scan association (:452-461) */
    /* S: disparity (valid-masked when invalid >= 0, :382-389) */
    for (int i = tid; i < P2; i += PREP_THREADS) {
        float x = 0.0f;
        if (i < H) {
            const float d = s_d[i];
            if (P.invalid >= 0) {
                const int va = d != P.invalid;
                x = ((float)va) * d;
            } else {
                x = d;
            }
        }
        s_pyr[i] = x;
    }
    __syncthreads();
    blelloch_build(s_pyr, P2, P.log2P2);
    float* svcol = sv_arr + (size_t)colg * 2 * (H + 1); /* compact copies for the pairwise phase 2 */
    for (int v = tid; v <= H; v += PREP_THREADS) {
        const float x = blelloch_prefix(s_pyr, P2, P.log2P2, v);
        rcol[v].S = x;
        svcol[v] = x;
    }
    __syncthreads();
    /* V: valid count (all zero without an invalid-disparity value: no scan needed) */
    if (P.invalid >= 0) {
        for (int i = tid; i < P2; i += PREP_THREADS)
            s_pyr[i] = (i < H) ? (float)(s_d[i] != P.invalid) : 0.0f;
        __syncthreads();
        blelloch_build(s_pyr, P2, P.log2P2);
        for (int v = tid; v <= H; v += PREP_THREADS) {
            const float x = blelloch_prefix(s_pyr, P2, P.log2P2, v);
            rcol[v].V = x;
            svcol[H + 1 + v] = x;
        }
        __syncthreads();
    } else {
        for (int v = tid; v <= H; v += PREP_THREADS) {
            rcol[v].V = 0.0f;
            svcol[H + 1 + v] = 0.0f;
        }
    }
    /* G: ground data cost, +inf at / above the horizon (:435-446) */
    for (int i = tid; i < P2; i += PREP_THREADS) {
        float x = 0.0f;
        if (i < H)
            x = (i >= vhor) ? IS_INF : data_cost_ground(gfun[i], s_d[i], gnorm[i], gis2[i], P);
        s_pyr[i] = x;
    }
    __syncthreads();
    blelloch_build(s_pyr, P2, P.log2P2);
    for (int v = tid; v <= H; v += PREP_THREADS) rcol[v].G = blelloch_prefix(s_pyr, P2, P.log2P2, v);
    __syncthreads();
    /* K: sky data cost, 0 below the horizon (:424-433) */
    for (int i = tid; i < P2; i += PREP_THREADS) {
        float x = 0.0f;
        if (i < H) x = (i < vhor) ? 0.0f : data_cost_sky(s_d[i], P);
        s_pyr[i] = x;
    }
    __syncthreads();
    blelloch_build(s_pyr, P2, P.log2P2);
    for (int v = tid; v <= H; v += PREP_THREADS) rcol[v].K = blelloch_prefix(s_pyr, P2, P.log2P2, v);
}

/* ====================================================================================== */
/* A4  object data-cost prefix table (ComputeObjectLUT, StixelsKernels.cu:236-296, 959-978) */
/* ====================================================================================== */
/* lutT[v][fn] = prefix over rows of obj_cost_lut[fn][(int)d[row]] with the reference's
 * association: per 32-row block a 32-lane Kogge-Stone network (shuffle distances 1,2,4,8,16)
 * whose lane 0 first receives the running carry; blocks chained serially.
 *
 * The reference gives a warp one fn and lets lanes be rows (5 shuffles per block).  Here a LANE
 * owns one fn and evaluates the same 32-input network on registers (129 fp32 adds per block, no
 * cross-lane traffic); the 64 lanes of a wave are 64 consecutive fn, so every load of the
 * transposed cost table and every store of a lutT row is one fully coalesced 256-byte access. */
#define LUT_BLOCK 32
__global__ __launch_bounds__(64) void k_object_lut(const DevParams P,
                                                   const float* __restrict__ joined,
                                                   const float* __restrict__ cost_T /*[dis][fn]*/,
                                                   float* __restrict__ lutT) {
    const int H = P.H, D = P.D;
    const int colg = blockIdx.x;
    const int lane = threadIdx.x;
    const int fn = blockIdx.y * 64 + lane;
    const bool fn_ok = fn < D;
    const int fnc = fn_ok ? fn : D - 1;
    const float* dcol = joined + (size_t)colg * H;
    float* lcol = lutT + (size_t)colg * (H + 1) * D;
    if (fn_ok) lcol[fn] = 0.0f; /* arr[0] = 0, :283-285 */
    float add = 0.0f;
    /* the per-row costs of the NEXT block are fetched before this block's 32 row stores are issued:
     * a wave's memory operations retire in order, so loads queued behind the stores would wait
     * for the whole store latency */
    float cn[LUT_BLOCK];
    auto fetch = [&](int i, float (&c)[LUT_BLOCK]) {
        /* (int)d of the block's 32 rows: one coalesced load, then wave-uniform broadcasts */
        const int rl = i + (lane & (LUT_BLOCK - 1));
        int dis_l = 0; /* rows beyond the image use dis = 0, :244-247 */
        if (rl < H) dis_l = (int)dcol[rl];
        dis_l = min(max(dis_l, 0), D - 1); /* memory safety outside the input domain (Q8) */
#pragma unroll
        for (int l = 0; l < LUT_BLOCK; l++) {
            const int dis = __builtin_amdgcn_readlane(dis_l, l);
            c[l] = cost_T[(size_t)dis * D + fnc];
        }
    };
    fetch(0, cn);
    for (int i = 0; i < H; i += LUT_BLOCK) {
        float c[LUT_BLOCK];
#pragma unroll
        for (int l = 0; l < LUT_BLOCK; l++) c[l] = cn[l];
        if (i + LUT_BLOCK < H) fetch(i + LUT_BLOCK, cn);
        c[0] += add; /* :249-251 */
#pragma unroll
        for (int j = 1; j < LUT_BLOCK; j <<= 1) { /* :255-263; descending l reads pre-step values */
#pragma unroll
            for (int l = LUT_BLOCK - 1; l >= j; l--) c[l] += c[l - j];
        }
        if (fn_ok) {
#pragma unroll
            for (int l = 0; l < LUT_BLOCK; l++)
                if (i + l < H) lcol[(size_t)(i + l + 1) * D + fn] = c[l]; /* :266 */
        }
        add = c[LUT_BLOCK - 1]; /* :268-272 */
    }
}

/* ====================================================================================== */
/* Pairwise transition priors that depend only on vB and the frame's ground model          */
/* ====================================================================================== */
__device__ __forceinline__ float neg_fastlog_div(float v, float v2) { /* :35-38 */
    return -is_logf(v) + is_logf(v2);
}

__global__ void k_prior_tables(const DevParams P, const float* __restrict__ ground,
                               PriorRec* __restrict__ priors, int n_images) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_images * P.H) return;
    const int img = idx / P.H, vB = idx % P.H;
    const float* gfun = ground + (size_t)img * 3 * P.H;
    PriorRec r;
    r.pc = neg_fastlog_div(1.0f, (float)(P.H - vB));
    r.g_from = P.nlog03 + r.pc;
    float gprev = (vB > 0) ? gfun[vB - 1] : 0.0f;
    r.s_from_g = (gprev < 1.0f) ? r.pc : IS_INF;
    r.o_from_s = neg_fastlog_div(1.0f, P.max_disf - P.epsilon) + r.pc;
    const float base = P.nlog07 + r.pc;
    if (gprev < 0.0f) gprev = 0.0f;
    r.g_prev = gprev;
    r.og_hi = base + neg_fastlog_div(P.pgrav, P.max_disf - gprev - P.epsilon);
    r.og_lo = base + neg_fastlog_div(P.pblg, gprev - P.epsilon);
    r.og_mid = base + neg_fastlog_div(1.0f - P.pgrav - P.pblg, 2.0f * P.epsilon);
    priors[idx] = r;
}

/* ====================================================================================== */
/* Segment evaluation shared by both DP kernels                                            */
/* ====================================================================================== */
struct SegTerms {
    float seg_g, seg_o, seg_s; /* semantic + instance terms of the three geometric classes */
    float gd, sd;              /* ground / sky data terms                                   */
    float mean;                /* un-floored, clamped (>= 0) object mean disparity          */
    int fni;                   /* floor(mean), clamped to [0, D-1]                          */
};

/* v_cvt_u32_f32: round toward zero, saturating (negative -> 0, NaN -> 0) */
__device__ __forceinline__ unsigned cvt_u32_sat(float x) {
    unsigned u;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(u) : "v"(x));
    return u;
}

/* a / h for an integer-valued h in [1, 11000] with r = RN(1/h): one multiplication and two
 * FMAs give the correctly rounded IEEE quotient for every fp32 a in [2^-100, 2^100] and a = 0
 * (exhaustively verified over all 2^23 mantissas x all h by tools/verify_exact_division.c). */
__device__ __forceinline__ float fast_div(float a, float h, float r) {
    const float q0 = a * r;
    const float e0 = __builtin_fmaf(-q0, h, a);
    return __builtin_fmaf(e0, r, q0);
}

/* wave-uniform record through the constant address space: scalar loads, values live in SGPRs */
__device__ __forceinline__ RowRec sload_rec(const RowRec* p) {
    crec_t q = (crec_t)p;
    RowRec r;
    r.Fg0 = q->Fg0; r.Fg1 = q->Fg1;
#pragma unroll
    for (int c = 0; c < IS_N_ON; c++) r.Fon[c] = q->Fon[c];
#pragma unroll
    for (int c = 0; c < IS_N_OI; c++) r.Foi[c] = q->Foi[c];
    r.Fsky = q->Fsky; r.Fnic = q->Fnic;
    r.G = q->G; r.K = q->K; r.S = q->S; r.V = q->V;
    r.MX = q->MX; r.MY = q->MY; r.MX2h = q->MX2h; r.MX2l = q->MX2l;
    r.MY2h = q->MY2h; r.MY2l = q->MY2l; r.pad[0] = q->pad[0]; r.pad[1] = q->pad[1];
    return r;
}

/* `my` = record at vT+1 (per lane), `rb` = record at vB (wave-uniform copy in SGPRs), r = RN(1/h).
 * Exact rewrites w.r.t. Cityscapes.h:44-118 / StixelsKernels.cu:62-86:
 *   DownsampledSum(c) = my.F_c - rb.F_c;
 *   min_c (k + float(S_c)) = k + float(min_c S_c) for classes sharing the additive term k
 *   (int -> float conversion and fp32 addition are monotone);
 *   (0.0f + k) + float(S) = k + float(S): float(int) is never -0, so the leading `0.0f +` of
 *   Cityscapes.h:67-79 cannot change the sum;
 *   FAST columns only: float(int64 difference) via exact fp32 hi/lo parts (RowRec), x / h via
 *   fast_div, and
 *   (int)floorf(max(mean,0)) = (int)fmaxf(mean,0) because the mean is finite there. */
template <bool FAST, bool HAS_INVALID>
__device__ __forceinline__ SegTerms eval_segment(const RowRec& my, const RowRec& rb, float height,
                                                 float r, int D, float iw) {
    SegTerms t;
    const float nic = iw * (float)(my.Fnic - rb.Fnic); /* ComputeNonInstanceOffsetCost, :62-70, :496-499 */
    float ic; /* ComputeInstanceOffsetCost, :72-86 */
    float f_g, f_on, f_oi, f_sky; /* float(min_c DownsampledSum_c) per class group */
    if (FAST) {
        f_g = __builtin_fminf(my.Fg0 - rb.Fg0, my.Fg1 - rb.Fg1);
        f_on = my.Fon[0] - rb.Fon[0];
#pragma unroll
        for (int c = 1; c < IS_N_ON; c++) f_on = __builtin_fminf(f_on, my.Fon[c] - rb.Fon[c]);
        f_oi = my.Foi[0] - rb.Foi[0];
#pragma unroll
        for (int c = 1; c < IS_N_OI; c++) f_oi = __builtin_fminf(f_oi, my.Foi[c] - rb.Foi[c]);
        f_sky = my.Fsky - rb.Fsky;
        const float meanx = my.MX - rb.MX;
        const float meany = my.MY - rb.MY;
        const float meanx2 = (my.MX2h - rb.MX2h) + (my.MX2l - rb.MX2l);
        const float meany2 = (my.MY2h - rb.MY2h) + (my.MY2l - rb.MY2l);
        ic = iw * (meanx2 - fast_div(meanx * meanx, height, r) + meany2 -
                   fast_div(meany * meany, height, r));
    } else {
        const RowRecWide& mw = reinterpret_cast<const RowRecWide&>(my);
        const RowRecWide& bw = reinterpret_cast<const RowRecWide&>(rb);
        f_g = (float)min(mw.Fg0 - bw.Fg0, mw.Fg1 - bw.Fg1);
        int32_t s_on = mw.Fon[0] - bw.Fon[0];
#pragma unroll
        for (int c = 1; c < IS_N_ON; c++) s_on = min(s_on, mw.Fon[c] - bw.Fon[c]);
        int32_t s_oi = mw.Foi[0] - bw.Foi[0];
#pragma unroll
        for (int c = 1; c < IS_N_OI; c++) s_oi = min(s_oi, mw.Foi[c] - bw.Foi[c]);
        f_on = (float)s_on;
        f_oi = (float)s_oi;
        f_sky = (float)(mw.Fsky - bw.Fsky);
        const float meanx = (float)(mw.MX - bw.MX);
        const float meany = (float)(mw.MY - bw.MY);
        const float meanx2 = (float)(mw.MX2 - bw.MX2);
        const float meany2 = (float)(mw.MY2 - bw.MY2);
        ic = iw * (meanx2 - meanx * meanx / height + meany2 - meany * meany / height);
    }

    t.seg_g = f_g + nic;
    const float on = nic + f_on;
    const float oi = ic + f_oi;
    t.seg_o = FAST ? __builtin_fminf(oi, on) : ((oi < on) ? oi : on); /* both finite when FAST */
    t.seg_s = f_sky + nic;

    t.gd = my.G - rb.G;
    t.sd = my.K - rb.K;
    float mean; /* ComputeMean, :47-60 */
    if (HAS_INVALID) {
        const float valid_dif = my.V - rb.V;
        mean = (valid_dif == 0) ? 0 : (my.S - rb.S) / valid_dif;
    } else if (FAST) {
        mean = fast_div(my.S - rb.S, height, r);
    } else {
        mean = (my.S - rb.S) / height;
    }
    if (FAST) {
        /* :525-527; the mean is finite in FAST columns, and v_cvt_u32_f32 saturates (x < 0 -> 0),
         * so the clamp at 0 is part of the conversion; = floorf for a finite mean >= 0 */
        t.fni = (int)min(cvt_u32_sat(mean), (unsigned)(D - 1));
        t.mean = __builtin_fmaxf(mean, 0.0f); /* (only the pairwise model reads it) */
    } else {
        if (mean < 0) mean = 0; /* :525-527 */
        t.mean = mean;
        const int fni = (int)__builtin_floorf(mean);
        t.fni = min(max(fni, 0), D - 1); /* memory safety outside the input domain (Q8) */
    }
    return t;
}

__device__ __forceinline__ RowRec load_rec(const RowRec* p) {
    RowRec r;
    const int4* s = (const int4*)p;
    int4* d = (int4*)&r;
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = s[i];
    return r;
}

/* lutT rows tile_lo+1 .. tile_lo+64 of a column into the LDS tile (row stride D+1).  When the
 * workgroup covers whole rows per sweep (nthreads a multiple of D) a thread keeps its column and
 * walks the rows: no per-element division.  QUADS: 16-byte loads (needs D % 4 == 0 and
 * 4 * nthreads a multiple of D); more registers, so the unary kernel (64 VGPRs, four loop nests)
 * uses the dword form. */
template <bool QUADS>
__device__ __forceinline__ void stage_lut_tile(float* s_tile, const float* __restrict__ lcol, int tile_lo,
                                               int H, int D, int tid, int nthreads) {
    const int DP = D + 1;
    if (QUADS && (D & 3) == 0 && ((4 * nthreads) % D) == 0) {
        const int quads = D >> 2;            /* 16-byte chunks per row */
        const int r0 = tid / quads, f = (tid - r0 * quads) * 4;
        const int dr = nthreads / quads;     /* rows per sweep of the workgroup */
        for (int r = r0; r < IS_TILE; r += dr) {
            const int v = min(tile_lo + 1 + r, H);
            const float4 x = *reinterpret_cast<const float4*>(lcol + (size_t)v * D + f);
            float* d = s_tile + r * DP + f;
            d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
        }
    } else if ((nthreads % D) == 0) {
        const int r0 = tid / D, f = tid - r0 * D;
        const int dr = nthreads / D;
        for (int r = r0; r < IS_TILE; r += dr) {
            const int v = min(tile_lo + 1 + r, H);
            s_tile[r * DP + f] = lcol[(size_t)v * D + f];
        }
    } else {
        for (int i = tid; i < IS_TILE * D; i += nthreads) {
            const int r = i / D, f = i - r * D;
            const int v = min(tile_lo + 1 + r, H);
            s_tile[r * DP + f] = lcol[(size_t)v * D + f];
        }
    }
}

/* ====================================================================================== */
/* A7-A9  unary DP: one workgroup = (column, 64-row tile)                                  */
/* ====================================================================================== */
/* In unary mode the predecessor cost is never added: cost_table[vT][t] is the
 * minimum over vB of a single-segment cost, so all (vB, vT) pairs are independent and the only
 * order that matters is the strict-< tie rule (smallest vB wins).  index_table holds the winning
 * vB (or -1); the predecessor TYPE is resolved in k_backtrace from the final cost_table. */
struct UnaryBest {
    float g, o, s;
    int vg, vo, vs;
};

/* One (vB, vT) evaluation of the unary model.
 *   SKY   the segment's lower neighbour is at / above the horizon (vB-1 >= vhor, :729): the
 *         non-object candidate is SKY, otherwise GROUND (:687); wave-uniform, so the caller runs
 *         separate loops and each loop has ONE accumulator pair live in its body;
 *   DIAG  the segment start may lie above this lane's vT (diagonal 64x64 block): masked lanes;
 *   FIRST vB = 0: ground additionally needs vT <= vhor (:542-545). */
/* vB-side row of lutT.  NR > 0: the wave holds the whole row in NR registers per lane (element
 * j*64 + lane), fetched with coalesced loads one step AHEAD of its use -- the address does not
 * depend on the segment -- and a lane picks its element fni with ds_bpermute (no memory access on
 * the dependent chain mean -> fni -> LUT value).  NR == 0 (D > 64*NR_MAX): per-lane gather. */
template <int NR>
struct LutRow {
    float r[NR > 0 ? NR : 1];
    const float* lrow;
};
/* Whole-row fetch through a raw buffer resource of the column's lutT: scalar row offset (SALU),
 * lane offset in a VGPR, no VALU address arithmetic; reads past the column return 0. */
template <int NR>
__device__ __forceinline__ void load_lut_row(LutRow<NR>& row, __amdgpu_buffer_rsrc_t lrsrc,
                                             const float* __restrict__ lcol, int v, int D, int lane4) {
    row.lrow = lcol + (size_t)v * D;
#pragma unroll
    for (int j = 0; j < NR; j++)
        row.r[j] = __int_as_float(
            __builtin_amdgcn_raw_buffer_load_b32(lrsrc, lane4, v * D * 4 + j * 256, 0));
}
template <int NR>
__device__ __forceinline__ float pick_lut(const LutRow<NR>& row, int fni) {
    if (NR == 0) return row.lrow[(unsigned)fni];
    const int sel = fni << 2; /* ds_bpermute takes the source lane from address bits [7:2] */
    float v = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(row.r[0])));
#pragma unroll
    for (int j = 1; j < NR; j++) {
        const float vj = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(row.r[j])));
        v = (fni >= 64 * j) ? vj : v;
    }
    return v;
}

/* (best, best_v) <- (cost, vB) in the lanes with cost < best: v_cmpx + two moves under the
 * resulting EXEC (8 issue cycles) instead of compare + two cndmask + the broadcast of vB (14).
 * Only for steps in which every lane of the wave takes part. */
__device__ __forceinline__ void take_if_less(float& best, int& best_v, float cost, int vB) {
    unsigned long long saved;
    asm("s_mov_b64 %[sv], exec\n\t"
        "v_cmpx_lt_f32_e32 %[c], %[b]\n\t"
        "v_mov_b32_e32 %[b], %[c]\n\t"
        "v_mov_b32_e32 %[i], %[vb]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [b] "+v"(best), [i] "+v"(best_v), [sv] "=&s"(saved)
        : [c] "v"(cost), [vb] "s"(vB)
        : "vcc");
}

/* same with a per-lane value to record */
__device__ __forceinline__ void take_if_less_v(float& best, int& best_v, float cost, int v) {
    unsigned long long saved;
    asm("s_mov_b64 %[sv], exec\n\t"
        "v_cmpx_lt_f32_e32 %[c], %[b]\n\t"
        "v_mov_b32_e32 %[b], %[c]\n\t"
        "v_mov_b32_e32 %[i], %[v]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [b] "+v"(best), [i] "+v"(best_v), [sv] "=&s"(saved)
        : [c] "v"(cost), [v] "v"(v)
        : "vcc");
}

template <bool FAST, bool HAS_INVALID, bool SKY, bool DIAG, bool FIRST, int NR, bool NOGROUND = false>
__device__ __forceinline__ void unary_step(const DevParams& P, const RowRec& my, const RowRec& rb,
                                           const LutRow<NR>& lrow, const float* my_tile,
                                           const float* s_rcp, int vT, int vTc, int vhor, int vB,
                                           float hf_full, bool row_ok, UnaryBest& b) {
    const int h = vTc + 1 - vB;
    const bool live = DIAG ? ((h > 0) && row_ok) : row_ok;
    const int hc = DIAG ? max(h, 1) : h;
    const float r = s_rcp[hc]; /* RN(1/h) = (float)(1./h) = inverse_height, :485, :608 */
    /* full steps: the caller carries the height as a float (one 2-cycle subtraction per step
     * instead of an integer update plus a conversion; integers < 2^24 are exact) */
    const float hf = (DIAG || FIRST) ? (float)hc : hf_full;
    const SegTerms t = eval_segment<FAST, HAS_INVALID>(my, rb, hf, r, P.D, P.iw);
    const float od = my_tile[t.fni] - pick_lut<NR>(lrow, t.fni);
    const float pwih = P.pw * r;
    /* cost = dw*data + pw*(1/h) + sw*seg, left to right (:716-719, 762-765, 820-823) */
    const float cost_o = P.dw * od + pwih + P.sw * t.seg_o;
    /* full steps: every lane with vT < H is live, and rows vT >= H are never stored */
    constexpr bool ALL_LANES = IS_CMPX_UPDATE && FAST && !DIAG && !FIRST;
    if (ALL_LANES) {
        take_if_less(b.o, b.vo, cost_o, vB);
    } else {
        const bool uo = live && (cost_o < b.o);
        b.o = uo ? cost_o : b.o;
        b.vo = uo ? vB : b.vo;
    }
    if (SKY) {
        const float cost_s = P.dw * t.sd + pwih + P.sw * t.seg_s;
        if (ALL_LANES) {
            take_if_less(b.s, b.vs, cost_s, vB);
        } else {
            const bool us = live && (cost_s < b.s);
            b.s = us ? cost_s : b.s;
            b.vs = us ? vB : b.vs;
        }
    } else if (!NOGROUND) {
        /* NOGROUND: every lane of the tile lies at or above the horizon, where the ground data
         * cost prefix is +inf (StixelsKernels.cu:435-446): dw * inf is inf (or NaN for dw = 0)
         * and never passes the strict < test, so the candidate is not evaluated at all */
        const float cost_g = P.dw * t.gd + pwih + P.sw * t.seg_g;
        if (ALL_LANES) {
            take_if_less(b.g, b.vg, cost_g, vB);
        } else {
            const bool ug = (FIRST ? (live && (vT <= vhor)) : live) && (cost_g < b.g);
            b.g = ug ? cost_g : b.g;
            b.vg = ug ? vB : b.vg;
        }
    }
}

template <bool FAST, bool HAS_INVALID, bool SKY, bool DIAG, int NR, bool NOGROUND = false>
__device__ __forceinline__ int unary_range(const DevParams& P, const RowRec& my,
                                           const RowRec* __restrict__ rcol,
                                           const float* __restrict__ lcol, const float* my_tile,
                                           const float* s_rcp, int vT, int vTc, int vhor, int vB,
                                           int nw, int bound, bool row_ok, int lane4,
                                           __amdgpu_buffer_rsrc_t lrsrc, LutRow<NR>& next_row,
                                           UnaryBest& b) {
    float hf = (float)(vTc + 1 - vB);
    const float nwf = (float)nw;
    for (; vB <= bound; vB += nw) {
        const RowRec cur = sload_rec(rcol + vB);
        const LutRow<NR> row = next_row;
        load_lut_row<NR>(next_row, lrsrc, lcol, min(vB + nw, P.H), P.D, lane4); /* row H exists */
        unary_step<FAST, HAS_INVALID, SKY, DIAG, false, NR, NOGROUND>(P, my, cur, row, my_tile, s_rcp, vT,
                                                                      vTc, vhor, vB, hf, row_ok, b);
        hf -= nwf;
    }
    return vB;
}

/* The wave walks vB = w, w+nw, ... <= vB_end in ascending order through (at most) four ranges:
 * ground/full, ground/diagonal, sky/full, sky/diagonal (ground while vB <= vhor; "full" while
 * every lane of the tile has vT >= vB, i.e. vB <= tile_lo). */
template <bool FAST, bool HAS_INVALID, int NR>
__device__ __forceinline__ void unary_loop(const DevParams& P, const RowRec& my,
                                           const RowRec* __restrict__ rcol,
                                           const float* __restrict__ lcol, const float* my_tile,
                                           const float* s_rcp, int vT, int vTc, int vhor, int w,
                                           int nw, int tile_lo, int vB_end, int lane4,
                                           __amdgpu_buffer_rsrc_t lrsrc, UnaryBest& b) {
    const bool row_ok = vT < P.H;
    int vB = w;
    if (vB > vB_end) return;
    LutRow<NR> next_row; /* lutT row of the step that comes next */
    load_lut_row<NR>(next_row, lrsrc, lcol, vB, P.D, lane4);
    if (vB == 0) { /* first segment (:481-594): ground + object */
        const RowRec cur = sload_rec(rcol);
        const LutRow<NR> row = next_row;
        load_lut_row<NR>(next_row, lrsrc, lcol, min(nw, P.H), P.D, lane4);
        if (tile_lo == 0)
            unary_step<FAST, HAS_INVALID, false, true, true, NR>(P, my, cur, row, my_tile, s_rcp, vT,
                                                                 vTc, vhor, 0, 0.0f, row_ok, b);
        else
            unary_step<FAST, HAS_INVALID, false, false, true, NR>(P, my, cur, row, my_tile, s_rcp, vT,
                                                                  vTc, vhor, 0, 0.0f, row_ok, b);
        vB += nw;
    }
    if (IS_SKIP_GROUND_ABOVE_HORIZON && tile_lo >= vhor) /* whole tile at / above the horizon */
        vB = unary_range<FAST, HAS_INVALID, false, false, NR, true>(
            P, my, rcol, lcol, my_tile, s_rcp, vT, vTc, vhor, vB, nw, min(min(vhor, tile_lo), vB_end),
            row_ok, lane4, lrsrc, next_row, b);
    else
        vB = unary_range<FAST, HAS_INVALID, false, false, NR>(P, my, rcol, lcol, my_tile, s_rcp, vT, vTc,
                                                              vhor, vB, nw, min(min(vhor, tile_lo), vB_end),
                                                              row_ok, lane4, lrsrc, next_row, b);
    vB = unary_range<FAST, HAS_INVALID, false, true, NR>(P, my, rcol, lcol, my_tile, s_rcp, vT, vTc,
                                                         vhor, vB, nw, min(vhor, vB_end), row_ok, lane4,
                                                         lrsrc, next_row, b);
    vB = unary_range<FAST, HAS_INVALID, true, false, NR>(P, my, rcol, lcol, my_tile, s_rcp, vT, vTc,
                                                         vhor, vB, nw, min(tile_lo, vB_end), row_ok,
                                                         lane4, lrsrc, next_row, b);
    unary_range<FAST, HAS_INVALID, true, true, NR>(P, my, rcol, lcol, my_tile, s_rcp, vT, vTc, vhor, vB,
                                                   nw, vB_end, row_ok, lane4, lrsrc, next_row, b);
}

/* FASTCOLS: the launch handles only the columns of that encoding (col_flags), workgroups of the
 * other kind leave at once.  Two lean kernels instead of one that carries both loop nests: no
 * register spills, and the generic launch costs ~nothing when every column is FAST. */
template <bool HAS_INVALID, int NR, bool FASTCOLS>
__global__ __launch_bounds__(IS_UNARY_WAVES * 64, IS_UNARY_OCC) void k_dp_unary(const DevParams P, int ncols,
                                                  const RowRec* __restrict__ recs,
                                                  const float* __restrict__ lutT,
                                                  const float* __restrict__ rcp,
                                                  const int* __restrict__ vhor_arr,
                                                  const int* __restrict__ col_flags,
                                                  float* __restrict__ cost_table,
                                                  int32_t* __restrict__ index_table,
                                                  int pairs_per_wg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = P.H, D = P.D;
    const int DP = D + 1; /* padded row: conflict-free when lanes share fni */
    float* s_rcp = (float*)smem;              /* [H+1 -> x4] RN(1/h), kept across both tiles   */
    float* s_tile = s_rcp + ((H + 1 + 3) & ~3); /* [64][D+1] lutT rows tile_lo+1 .. tile_lo+64  */

    /* XCD-aware order: blocks b, b+8, ... share an XCD/L2; keep all tiles of a column on one
     * XCD (they gather from the same lutT) and start with the tallest tiles. */
    const int nxcd = 8;
    const int npairs = (P.ntiles + 1) / 2;
    const int wg_per_col = (npairs + pairs_per_wg - 1) / pairs_per_wg;
    /* (integer division runs on the VALU: pin the uniform results back into SGPRs) */
    const int xcd = blockIdx.x % nxcd, q = blockIdx.x / nxcd;
    const int wg_in_col = __builtin_amdgcn_readfirstlane(q % wg_per_col);
    const int colg = __builtin_amdgcn_readfirstlane((q / wg_per_col) * nxcd + xcd);
    if (colg >= ncols) return;
    if ((__builtin_amdgcn_readfirstlane(col_flags[colg]) == 0) != FASTCOLS) return;
    const int img = __builtin_amdgcn_readfirstlane(colg / P.C);
    const int vhor = __builtin_amdgcn_readfirstlane(vhor_arr[img]);

    const int tid = threadIdx.x, lane = tid & 63;
    const int nw = blockDim.x >> 6;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const RowRec* rcol = recs + (size_t)colg * (H + 1);
    const float* lcol = lutT + (size_t)colg * (H + 1) * D;
    for (int i = tid; i <= H; i += blockDim.x) s_rcp[i] = rcp[i];
    const __amdgpu_buffer_rsrc_t lrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)lcol, 0, (H + 1) * D * (int)sizeof(float), 0x00020000 /* raw, 32-bit data */);

    /* a workgroup takes the tile pair (ntiles-1-pair, pair): every workgroup then carries the
     * same number of (vB, vT) pairs, and the per-workgroup fixed costs are paid half as often */
    bool first = true;
    for (int pair = wg_in_col; pair < npairs; pair += wg_per_col) {
    const int n_pass = (P.ntiles - 1 - pair == pair) ? 1 : 2;
    for (int pass = 0; pass < n_pass; pass++) {
    const int tile = pass == 0 ? (P.ntiles - 1 - pair) : pair;
    const int tile_lo = tile * IS_TILE;
    if (!first) __syncthreads(); /* the merge area of the previous tile aliases the LUT tile */
    first = false;

    stage_lut_tile<false>(s_tile, lcol, tile_lo, H, D, tid, (int)blockDim.x);

    const int vT = tile_lo + lane;
    const int vTc = min(vT, H - 1);
    const RowRec my = load_rec(rcol + vTc + 1);
    __syncthreads();

    UnaryBest b;
    b.g = b.o = b.s = IS_INF;
    b.vg = b.vs = -1;
    b.vo = 0; /* index_table[vT*3+OBJECT] = OBJECT at vB = 0, :592 */
    const float* my_tile = s_tile + lane * DP;
    const int vB_end = min(tile_lo + IS_TILE - 1, H - 1);
    unary_loop<FASTCOLS, HAS_INVALID, NR>(P, my, rcol, lcol, my_tile, s_rcp, vT, vTc, vhor, w, nw, tile_lo,
                                          vB_end, lane * 4, lrsrc, b);

    /* merge the waves' partial minima: min cost, ties -> smallest vB (= first strict minimum
     * of the reference's ascending-vB loop) */
    __syncthreads();
    float* m_cost = s_tile;                        /* [nw][3][64] (aliases the LUT tile) */
    int* m_vb = (int*)(m_cost + nw * 3 * 64);      /* [nw][3][64] */
    m_cost[(w * 3 + 0) * 64 + lane] = b.g; m_vb[(w * 3 + 0) * 64 + lane] = b.vg;
    m_cost[(w * 3 + 1) * 64 + lane] = b.o; m_vb[(w * 3 + 1) * 64 + lane] = b.vo;
    m_cost[(w * 3 + 2) * 64 + lane] = b.s; m_vb[(w * 3 + 2) * 64 + lane] = b.vs;
    __syncthreads();
    if (tid < 3 * 64) {
        const int type = tid >> 6;
        float c = m_cost[(0 * 3 + type) * 64 + lane];
        int vb = m_vb[(0 * 3 + type) * 64 + lane];
        for (int ww = 1; ww < nw; ww++) {
            const float c2 = m_cost[(ww * 3 + type) * 64 + lane];
            const int vb2 = m_vb[(ww * 3 + type) * 64 + lane];
            const bool take = (c2 < c) || (c2 == c && vb2 >= 0 && (vb < 0 || vb2 < vb));
            if (take) { c = c2; vb = vb2; }
        }
        /* final (cost, vB) of this type back to LDS: one wave then writes the three types of a
         * row as 12 contiguous bytes (a wave-wide contiguous 768-byte store) instead of three
         * waves writing every third dword */
        m_cost[type * 64 + lane] = c;
        m_vb[type * 64 + lane] = vb;
    }
    __syncthreads();
    if (w == 0 && vT < H) {
        const size_t o = ((size_t)colg * H + vT) * 3;
        float* cd = cost_table + o;
        int32_t* id = index_table + o;
        cd[0] = m_cost[0 * 64 + lane]; cd[1] = m_cost[1 * 64 + lane]; cd[2] = m_cost[2 * 64 + lane];
        id[0] = m_vb[0 * 64 + lane]; id[1] = m_vb[1 * 64 + lane]; id[2] = m_vb[2 * 64 + lane];
    }
    } /* pass */
    } /* pair */
}

/* ====================================================================================== */
/* A7-A9  pairwise DP                                                                      */
/* ====================================================================================== */
/* Everything of a transition INTO a segment starting at vB that does not depend on the lane:
 * the final costs of row vB-1 combined with the transition priors (StixelsKernels.cu:88-199,
 * 687-837).  Built once, when row vB-1 becomes final, from the row's costs, the winning
 * object chain (previous_mean) and the frame's PriorRec; read by every later segment through
 * scalar loads.  64 bytes. */
struct __attribute__((aligned(64))) StepRec {
    float pwmp;      /* pw * fminf(p1, p2) of the ground (vB-1 < vhor) or sky transition       */
    int idx_gs;      /* vB*3 + (p1 < p2 ? GROUND : OBJECT)                    :723-727, 769-773 */
    float g_hi_thr, g_lo_thr;   /* g_prev + epsilon, g_prev - epsilon                 :132, 136 */
    float p1_hi, p1_lo, p1_mid; /* cG + pw * GetPriorCostObjectFromGround, three cases :120-144 */
    float o_hi_thr, o_lo_thr;   /* pm + dif, pm - dif                                 :159, 163 */
    float p2_hi, p2_lo, p2_mid; /* cO + pw * GetPriorCostObjectFromObject, three cases :146-171 */
    float p3_yes, p3_no;        /* cS + pw * GetPriorCostObjectFromSky, fn > eps or not :173-183 */
    float pad0, pad1;
};
static_assert(sizeof(StepRec) == 64, "StepRec must be 64 bytes");
typedef const __attribute__((address_space(4))) StepRec* cstep_t;

struct StepVals { /* register copy of a StepRec, always passed by value */
    float pwmp;
    int idx_gs;
    float g_hi_thr, g_lo_thr, p1_hi, p1_lo, p1_mid, o_hi_thr, o_lo_thr, p2_hi, p2_lo, p2_mid, p3_yes,
        p3_no;
};

__device__ __forceinline__ void store_step(StepRec* dst, const StepVals v) {
    float4* d = reinterpret_cast<float4*>(dst);
    d[0] = make_float4(v.pwmp, __builtin_bit_cast(float, v.idx_gs), v.g_hi_thr, v.g_lo_thr);
    d[1] = make_float4(v.p1_hi, v.p1_lo, v.p1_mid, v.o_hi_thr);
    d[2] = make_float4(v.o_lo_thr, v.p2_hi, v.p2_lo, v.p2_mid);
    d[3] = make_float4(v.p3_yes, v.p3_no, 0.0f, 0.0f);
}

/* A scalar-loaded value made opaque to the optimiser: without this, LLVM rewrites the selects
 * between neighbouring record fields (p1_hi / p1_lo / p1_mid ...) into a per-lane indexed load
 * from a scratch copy of the record. */
__device__ __forceinline__ float opaque_s(float x) {
    asm volatile("" : "+s"(x));
    return x;
}

__device__ __forceinline__ StepVals sload_step(const StepRec* p) {
    cstep_t q = (cstep_t)p;
    StepVals r;
    r.pwmp = q->pwmp; r.idx_gs = q->idx_gs;
    r.g_hi_thr = q->g_hi_thr; r.g_lo_thr = q->g_lo_thr;
    r.p1_hi = opaque_s(q->p1_hi); r.p1_lo = opaque_s(q->p1_lo); r.p1_mid = opaque_s(q->p1_mid);
    r.o_hi_thr = q->o_hi_thr; r.o_lo_thr = q->o_lo_thr;
    r.p2_hi = opaque_s(q->p2_hi); r.p2_lo = opaque_s(q->p2_lo); r.p2_mid = opaque_s(q->p2_mid);
    r.p3_yes = opaque_s(q->p3_yes); r.p3_no = opaque_s(q->p3_no);
    return r;
}

__device__ __forceinline__ float readlane_f(float x, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l));
}

/* -is_logf(v) + is_logf(v2) (NegFastLogDiv, :35-38) for TWO argument pairs at once: the lower
 * half of the wave evaluates pair A, the upper half pair B, so the serial chain pays for one
 * logarithm instead of two.  v is a compile-time-like constant whose log is passed in. */
__device__ __forceinline__ void neg_fastlog_div2(float neg_log_va, float v2a, float neg_log_vb,
                                                 float v2b, const double* s_invc,
                                                 const double* s_logc, float* outa, float* outb) {
    const bool upper = threadIdx.x >= 32;
    const float arg = upper ? v2b : v2a;
    const float l = is_logf_t(arg, s_invc, s_logc);
    const float la = readlane_f(l, 0), lb = readlane_f(l, 32);
    *outa = neg_log_va + la;
    *outb = neg_log_vb + lb;
}

/* StepRec of vB = r + 1 from the final row r.  All inputs are wave-uniform; every lane computes
 * the same values.  s_S / s_V: the column's disparity / valid-count prefixes in LDS. */
template <bool HAS_INVALID>
__device__ __forceinline__ StepVals make_step(const DevParams& P, const float* s_S, const float* s_V,
                                             const float* s_odr, const double* s_invc,
                                             const double* s_logc, cprior_t pr, int vhor, int r,
                                             float cG, float cO, float cS, int obj_vB) {
    const int vB = r + 1;
    const float pw = P.pw;
    StepVals st;
    /* previous_mean = ComputeMean(previous_object_vB, previous_vT), :47-60, :675-685 */
    float pm;
    if (HAS_INVALID) {
        const float valid_dif = s_V[r + 1] - s_V[obj_vB];
        pm = (valid_dif == 0) ? 0 : (s_S[r + 1] - s_S[obj_vB]) / valid_dif;
    } else {
        pm = (s_S[r + 1] - s_S[obj_vB]) / (float)(r + 1 - obj_vB);
    }
    if (pm < 0) pm = 0;
    const float pc = pr->pc;

    if (r < vhor) { /* ground, :687-728 */
        const float prev_cost = pr->g_from;
        const float p1 = cG + pw * prev_cost;
        const float p2 = cO + pw * prev_cost;
        st.pwmp = pw * __builtin_fminf(p1, p2);
        st.idx_gs = vB * 3 + ((p1 < p2) ? IS_GROUND : IS_OBJECT);
    } else { /* sky, :729-775 */
        const float p1 = cG + pw * pr->s_from_g;
        const float so = (pm < P.epsilon) ? IS_INF : (P.log2c + pc); /* :88-96 */
        const float p2 = cO + pw * so;
        st.pwmp = pw * __builtin_fminf(p1, p2);
        st.idx_gs = vB * 3 + ((p1 < p2) ? IS_GROUND : IS_OBJECT);
    }
    /* object from ground, :120-144 */
    st.g_hi_thr = pr->g_prev + P.epsilon;
    st.g_lo_thr = pr->g_prev - P.epsilon;
    st.p1_hi = cG + pw * pr->og_hi;
    st.p1_lo = cG + pw * pr->og_lo;
    st.p1_mid = cG + pw * pr->og_mid;
    /* object from object, :146-171 */
    float base = (r < vhor) ? P.nlog07 : P.log2c;
    base += pc;
    int k = (int)pm;
    k = min(max(k, 0), P.D - 1);
    float dif = s_odr[k];
    if (dif < 0.0f) dif = 0.0f;
    st.o_hi_thr = pm + dif;
    st.o_lo_thr = pm - dif;
    float nl_hi, nl_lo;
    neg_fastlog_div2(P.nlog_pord, P.max_disf - pm - dif, P.nlog_1mpord, st.o_lo_thr, s_invc, s_logc,
                     &nl_hi, &nl_lo);
    st.p2_hi = cO + pw * (base + nl_hi);
    st.p2_lo = cO + pw * (base + nl_lo);
    st.p2_mid = cO + pw * IS_INF;
    /* object from sky, :173-183 */
    st.p3_yes = cS + pw * pr->o_from_s;
    st.p3_no = cS + pw * IS_INF;
    return st;
}

struct PairBest {
    float g, o, s;
    int ig, io, is; /* vB*3 + prev type */
};

/* One (vB >= 1, vT) evaluation of the pairwise model for the lane owning vT; `st` is the
 * wave-uniform StepRec of vB.  SKY: vB-1 >= vhor (:729), else ground (:687). */
template <bool SKY, bool ALL_LANES = false, bool NOGROUND = false>
__device__ __forceinline__ void pairwise_step(const DevParams& P, const StepVals st, int vB,
                                              bool live, float od, const SegTerms& t, PairBest& b) {
    /* ALL_LANES (phase 1): every lane with vT < H is live and rows vT >= H are never stored */
    constexpr bool CMPX = IS_CMPX_UPDATE && ALL_LANES;
    if (SKY) { /* :729-775 */
        const float cost = P.dw * t.sd + st.pwmp + P.sw * t.seg_s;
        if (CMPX) {
            take_if_less(b.s, b.is, cost, st.idx_gs);
        } else {
            const bool u = live && (cost < b.s);
            b.s = u ? cost : b.s;
            b.is = u ? st.idx_gs : b.is;
        }
    } else if (!NOGROUND) { /* :687-728; NOGROUND: tile at / above the horizon, see unary_step */
        const float cost = P.dw * t.gd + st.pwmp + P.sw * t.seg_g;
        if (CMPX) {
            take_if_less(b.g, b.ig, cost, st.idx_gs);
        } else {
            const bool u = live && (cost < b.g);
            b.g = u ? cost : b.g;
            b.ig = u ? st.idx_gs : b.ig;
        }
    }
    /* object, :777-837 */
    const float fn = t.mean;
    const float p1 = (fn > st.g_hi_thr) ? st.p1_hi : ((fn < st.g_lo_thr) ? st.p1_lo : st.p1_mid);
    const float p2 = (fn > st.o_hi_thr) ? st.p2_hi : ((fn < st.o_lo_thr) ? st.p2_lo : st.p2_mid);
    const float p3 = (fn > P.epsilon) ? st.p3_yes : st.p3_no;
    const float m12 = __builtin_fminf(p1, p2);
    const float mp = __builtin_fminf(m12, p3);
    const float cost = P.dw * od + P.pw * mp + P.sw * t.seg_o;
    /* min_prev: OBJECT (1), GROUND (0) if p1 < p2, SKY (2) if p3 < fminf(p1, p2), :828-835 */
    const int base_o = vB * 3 + IS_OBJECT;
    int idx = (p1 < p2) ? (base_o - 1) : base_o;
    idx = (p3 < m12) ? (base_o + 1) : idx;
    if (CMPX) {
        take_if_less_v(b.o, b.io, cost, idx);
    } else {
        const bool u = live && (cost < b.o);
        b.o = u ? cost : b.o;
        b.io = u ? idx : b.io;
    }
}

/* The pairwise DP of one 64-row tile is split over two launches (per tile, bottom-up):
 *
 *  phase 1  k_pw_phase1: segments that START in earlier tiles (vB <= tile_lo).  Their
 *           predecessor rows are final (StepRec written by earlier launches, read with scalar
 *           loads), so all (vB, vT) pairs are independent: same structure, occupancy and issue
 *           bound as the unary kernel.  Writes the merged partial minima of the tile.
 *  phase 2  k_pw_phase2: the 64x64 diagonal block, where step vB needs the final row vB-1 of
 *           the same tile: one wavefront per column walks the 63 steps; the finished row is
 *           broadcast with v_readlane, its StepRec is computed uniformly and published.
 *
 * The serial chain of the reference (rows x __syncthreads, StixelsKernels.cu:600-603) is thus
 * confined to phase 2, 1/16 of the pair evaluations at 1024 rows. */
template <bool FAST, bool HAS_INVALID, int NR>
__device__ __forceinline__ void pw_phase1_body(const DevParams& P, char* smem, int colg, int tile,
                                               const RowRec* __restrict__ recs,
                                               const float* __restrict__ lutT,
                                               const StepRec* __restrict__ steps,
                                               const float* __restrict__ rcp, int vhor,
                                               int split, int nsplit,
                                               float* __restrict__ part_cost,
                                               int* __restrict__ part_idx) {
    const int H = P.H, D = P.D;
    const int DP = D + 1;
    float* s_tile = (float*)smem;             /* [64][D+1] */
    float* s_rcp = s_tile + IS_TILE * DP;     /* [H+1]     */
    const int tid = threadIdx.x, lane = tid & 63;
    /* `nsplit` workgroups share the vB range of one (column, tile): together they behave like
     * one workgroup of nsplit * nwl waves (few columns = small batches: more of the chip works
     * on the latency chain); their partial minima are merged by phase 2 */
    const int nwl = blockDim.x >> 6;
    const int wl = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = nwl * nsplit;
    const int w = split * nwl + wl;
    const int tile_lo = tile * IS_TILE;
    const RowRec* rcol = recs + (size_t)colg * (H + 1);
    const float* lcol = lutT + (size_t)colg * (H + 1) * D;
    const StepRec* scol = steps + (size_t)colg * H;

    stage_lut_tile<true>(s_tile, lcol, tile_lo, H, D, tid, (int)blockDim.x);
    for (int i = tid; i <= H; i += blockDim.x) s_rcp[i] = rcp[i];
    const int vT = tile_lo + lane;
    const int vTc = min(vT, H - 1);
    const RowRec my = load_rec(rcol + vTc + 1);
    const float* my_tile = s_tile + lane * DP;
    const bool live = vT < H;
    __syncthreads();

    PairBest b;
    b.g = b.o = b.s = IS_INF;
    b.ig = b.is = -1;
    b.io = IS_OBJECT; /* :592 */
    const int vB_last = min(tile_lo, H - 1);
    int vB = w;
    /* vB-side lutT row: fetched one step ahead, picked with ds_bpermute (see LutRow) */
    const __amdgpu_buffer_rsrc_t lrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)lcol, 0, (H + 1) * D * (int)sizeof(float), 0x00020000);
    const int lane4 = lane * 4;
    LutRow<NR> next_row;
    if (vB <= vB_last) load_lut_row<NR>(next_row, lrsrc, lcol, vB == 0 ? min(nw, H) : vB, D, lane4);
    if (vB == 0) { /* first segment, :481-594 */
        const RowRec rb = sload_rec(rcol);
        const int h = vTc + 1;
        const SegTerms t = eval_segment<FAST, HAS_INVALID>(my, rb, (float)h, s_rcp[h], D, P.iw);
        const float od = my_tile[t.fni] - lcol[(unsigned)t.fni];
        const bool below = vT <= vhor;
        const float cost_g = P.dw * t.gd + P.pw * P.first_g + P.sw * t.seg_g;
        const bool ug = live && below && (cost_g < b.g);
        b.g = ug ? cost_g : b.g;
        b.ig = ug ? IS_GROUND : b.ig;
        const float prior = below ? P.first_o_below : P.first_o_above;
        const float cost = P.dw * od + P.pw * prior + P.sw * t.seg_o;
        b.o = (live && cost < b.o) ? cost : b.o;
        vB += nw;
    }
    if (IS_SKIP_GROUND_ABOVE_HORIZON && tile_lo >= vhor) {
        for (; vB <= min(vhor, vB_last); vB += nw) { /* ground range, ground candidate = +inf */
            const RowRec rb = sload_rec(rcol + vB);
            const StepVals st = sload_step(scol + vB);
            const LutRow<NR> row = next_row;
            load_lut_row<NR>(next_row, lrsrc, lcol, min(vB + nw, H), D, lane4);
            const int h = vTc + 1 - vB;
            const SegTerms t = eval_segment<FAST, HAS_INVALID>(my, rb, (float)h, s_rcp[h], D, P.iw);
            const float od = my_tile[t.fni] - pick_lut<NR>(row, t.fni);
            pairwise_step<false, true, true>(P, st, vB, live, od, t, b);
        }
    }
    for (; vB <= min(vhor, vB_last); vB += nw) { /* ground range: vB-1 < vhor */
        const RowRec rb = sload_rec(rcol + vB);
        const StepVals st = sload_step(scol + vB);
        const LutRow<NR> row = next_row;
        load_lut_row<NR>(next_row, lrsrc, lcol, min(vB + nw, H), D, lane4);
        const int h = vTc + 1 - vB;
        const SegTerms t = eval_segment<FAST, HAS_INVALID>(my, rb, (float)h, s_rcp[h], D, P.iw);
        const float od = my_tile[t.fni] - pick_lut<NR>(row, t.fni);
        pairwise_step<false, true>(P, st, vB, live, od, t, b);
    }
    for (; vB <= vB_last; vB += nw) { /* sky range */
        const RowRec rb = sload_rec(rcol + vB);
        const StepVals st = sload_step(scol + vB);
        const LutRow<NR> row = next_row;
        load_lut_row<NR>(next_row, lrsrc, lcol, min(vB + nw, H), D, lane4);
        const int h = vTc + 1 - vB;
        const SegTerms t = eval_segment<FAST, HAS_INVALID>(my, rb, (float)h, s_rcp[h], D, P.iw);
        const float od = my_tile[t.fni] - pick_lut<NR>(row, t.fni);
        pairwise_step<true, true>(P, st, vB, live, od, t, b);
    }
    /* merge the waves: min cost, ties -> smallest vB (a finite cost always has a real index) */
    __syncthreads();
    float* m_cost = (float*)smem;               /* [nwl][3][64] (aliases the tile) */
    int* m_idx = (int*)(m_cost + nwl * 3 * 64); /* [nwl][3][64] */
    m_cost[(wl * 3 + 0) * 64 + lane] = b.g; m_idx[(wl * 3 + 0) * 64 + lane] = b.ig;
    m_cost[(wl * 3 + 1) * 64 + lane] = b.o; m_idx[(wl * 3 + 1) * 64 + lane] = b.io;
    m_cost[(wl * 3 + 2) * 64 + lane] = b.s; m_idx[(wl * 3 + 2) * 64 + lane] = b.is;
    __syncthreads();
    if (tid < 3 * 64) {
        const int type = tid >> 6;
        float c = m_cost[(0 * 3 + type) * 64 + lane];
        int ix = m_idx[(0 * 3 + type) * 64 + lane];
        for (int ww = 1; ww < nwl; ww++) {
            const float c2 = m_cost[(ww * 3 + type) * 64 + lane];
            const int ix2 = m_idx[(ww * 3 + type) * 64 + lane];
            const bool take = (c2 < c) || (c2 == c && c2 < IS_INF && (ix2 / 3) < (ix / 3));
            if (take) { c = c2; ix = ix2; }
        }
        const size_t o = (((size_t)colg * nsplit + split) * 3 + type) * 64 + lane;
        part_cost[o] = c;
        part_idx[o] = ix;
    }
}

template <bool HAS_INVALID, int NR>
__global__ __launch_bounds__(IS_UNARY_WAVES * 64, IS_UNARY_WAVES) void k_pw_phase1(
    const DevParams P, int col_base, int ncols, int tile, int nsplit,
    const RowRec* __restrict__ recs, const float* __restrict__ lutT,
    const StepRec* __restrict__ steps, const float* __restrict__ rcp,
    const int* __restrict__ vhor_arr, const int* __restrict__ col_flags,
    float* __restrict__ part_cost, int* __restrict__ part_idx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int colg = col_base + (int)(blockIdx.x / (unsigned)nsplit);
    const int split = __builtin_amdgcn_readfirstlane((int)(blockIdx.x % (unsigned)nsplit));
    if (colg >= ncols) return;
    const int vhor = __builtin_amdgcn_readfirstlane(vhor_arr[colg / P.C]);
    if (__builtin_amdgcn_readfirstlane(col_flags[colg]) == 0)
        pw_phase1_body<true, HAS_INVALID, NR>(P, smem, colg, tile, recs, lutT, steps, rcp, vhor, split,
                                              nsplit, part_cost, part_idx);
    else
        pw_phase1_body<false, HAS_INVALID, NR>(P, smem, colg, tile, recs, lutT, steps, rcp, vhor, split,
                                               nsplit, part_cost, part_idx);
}

template <bool FAST, bool HAS_INVALID>
__device__ __forceinline__ void pw_phase2_body(const DevParams& P, char* smem, int colg, int tile,
                                               const RowRec* __restrict__ recs,
                                               const float* __restrict__ lutT,
                                               const PriorRec* __restrict__ priors,
                                               const float* __restrict__ odr,
                                               const float* __restrict__ rcp,
                                               const float* __restrict__ sv_arr, int vhor,
                                               int nsplit, const float* __restrict__ part_cost,
                                               const int* __restrict__ part_idx,
                                               StepRec* __restrict__ steps,
                                               float* __restrict__ cost_table,
                                               int32_t* __restrict__ index_table) {
    const int H = P.H, D = P.D;
    const int lane = threadIdx.x;
    double* s_invc = (double*)smem;                    /* [32] */
    double* s_logc = s_invc + IS_LOG_TABLE_SIZE;       /* [32] */
    float* s_S = (float*)(s_logc + IS_LOG_TABLE_SIZE); /* [H+1] */
    float* s_V = s_S + (H + 1);                        /* [H+1] */
    float* s_odr = s_V + (H + 1);                      /* [D]   */
    const int tile_lo = tile * IS_TILE;
    const RowRec* rcol = recs + (size_t)colg * (H + 1);
    const float* lcol = lutT + (size_t)colg * (H + 1) * D;
    const PriorRec* pcol = priors + (size_t)(colg / P.C) * H;
    StepRec* scol = steps + (size_t)colg * H;
    const float* sv = sv_arr + (size_t)colg * 2 * (H + 1);
    if (lane == 0) is_log_tables(s_invc, s_logc);
    for (int i = lane; i <= H; i += 64) {
        s_S[i] = sv[i];
        if (HAS_INVALID) s_V[i] = sv[H + 1 + i];
    }
    for (int i = lane; i < D; i += 64) s_odr[i] = odr[i];
    const int vT = tile_lo + lane;
    const int vTc = min(vT, H - 1);
    const RowRec my = load_rec(rcol + vTc + 1);
    const float* my_row = lcol + (size_t)(vTc + 1) * D;
    PairBest b; /* partial minima of phase 1 (its nsplit workgroups merged: min cost, then smallest vB) */
    {
        const size_t o = (size_t)colg * nsplit * 3 * 64 + lane;
        b.g = part_cost[o]; b.ig = part_idx[o];
        b.o = part_cost[o + 64]; b.io = part_idx[o + 64];
        b.s = part_cost[o + 128]; b.is = part_idx[o + 128];
        for (int sp = 1; sp < nsplit; sp++) {
            const size_t q = o + (size_t)sp * 3 * 64;
            float c2 = part_cost[q]; int i2 = part_idx[q];
            if ((c2 < b.g) || (c2 == b.g && c2 < IS_INF && (i2 / 3) < (b.ig / 3))) { b.g = c2; b.ig = i2; }
            c2 = part_cost[q + 64]; i2 = part_idx[q + 64];
            if ((c2 < b.o) || (c2 == b.o && c2 < IS_INF && (i2 / 3) < (b.io / 3))) { b.o = c2; b.io = i2; }
            c2 = part_cost[q + 128]; i2 = part_idx[q + 128];
            if ((c2 < b.s) || (c2 == b.s && c2 < IS_INF && (i2 / 3) < (b.is / 3))) { b.s = c2; b.is = i2; }
        }
    }
    __syncthreads();

    const int n_rows = min(IS_TILE, H - tile_lo);
    StepVals st;
    st.pwmp = IS_INF; st.idx_gs = -1;
    st.g_hi_thr = st.g_lo_thr = st.p1_hi = st.p1_lo = st.p1_mid = st.o_hi_thr = st.o_lo_thr = 0.0f;
    st.p2_hi = st.p2_lo = st.p2_mid = st.p3_yes = st.p3_no = 0.0f;
    for (int s = 0; s < n_rows; s++) {
        const int r = tile_lo + s; /* row that becomes final in this step */
        if (s > 0) { /* segments starting at vB = r: lanes vT >= r */
            const RowRec rb = sload_rec(rcol + r);
            const int hc = max(vTc + 1 - r, 1);
            const bool live = (vT < H) && (vT >= r);
            const SegTerms t = eval_segment<FAST, HAS_INVALID>(my, rb, (float)hc, rcp[hc], D, P.iw);
            const float od = my_row[(unsigned)t.fni] - (lcol + (size_t)r * D)[(unsigned)t.fni];
            if (r - 1 < vhor)
                pairwise_step<false>(P, st, r, live, od, t, b);
            else
                pairwise_step<true>(P, st, r, live, od, t, b);
        }
        /* lane s holds the final values of row r: broadcast, derive the StepRec of vB = r+1 */
        if (r + 1 < H) {
            const float cG = readlane_f(b.g, s), cO = readlane_f(b.o, s), cS = readlane_f(b.s, s);
            const int io = __builtin_amdgcn_readlane(b.io, s);
            st = make_step<HAS_INVALID>(P, s_S, s_V, s_odr, s_invc, s_logc, (cprior_t)(pcol + r + 1), vhor,
                                        r, cG, cO, cS, io / 3);
            if (lane == 0) store_step(scol + r + 1, st);
        }
    }
    if (vT < H) {
        const size_t o = ((size_t)colg * H + vT) * 3;
        cost_table[o + 0] = b.g; cost_table[o + 1] = b.o; cost_table[o + 2] = b.s;
        index_table[o + 0] = b.ig; index_table[o + 1] = b.io; index_table[o + 2] = b.is;
    }
}

template <bool HAS_INVALID>
__global__ __launch_bounds__(64) void k_pw_phase2(
    const DevParams P, int col_base, int ncols, int tile, int nsplit,
    const RowRec* __restrict__ recs, const float* __restrict__ lutT,
    const PriorRec* __restrict__ priors,
    const float* __restrict__ odr, const float* __restrict__ rcp,
    const float* __restrict__ sv_arr, const int* __restrict__ vhor_arr,
    const int* __restrict__ col_flags, const float* __restrict__ part_cost,
    const int* __restrict__ part_idx, StepRec* __restrict__ steps, float* __restrict__ cost_table,
    int32_t* __restrict__ index_table) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int colg = col_base + blockIdx.x;
    if (colg >= ncols) return;
    const int vhor = __builtin_amdgcn_readfirstlane(vhor_arr[colg / P.C]);
    if (__builtin_amdgcn_readfirstlane(col_flags[colg]) == 0)
        pw_phase2_body<true, HAS_INVALID>(P, smem, colg, tile, recs, lutT, priors, odr, rcp, sv_arr, vhor,
                                          nsplit, part_cost, part_idx, steps, cost_table, index_table);
    else
        pw_phase2_body<false, HAS_INVALID>(P, smem, colg, tile, recs, lutT, priors, odr, rcp, sv_arr, vhor,
                                           nsplit, part_cost, part_idx, steps, cost_table, index_table);
}

/* ====================================================================================== */
/* A10  back-tracing, one wavefront per column                                             */
/* ====================================================================================== */
/* Everything of one Section that depends only on (vT, vB, type): StixelsKernels.cu:868-944. */
__device__ __forceinline__ is_section make_section(const DevParams& P, const RowRec* rcol, bool wide,
                                                   int vT, int vB, int type, float cost) {
    const RowRec a = load_rec(rcol + vT + 1);
    const RowRec bq = load_rec(rcol + vB);
    const RowRecWide& aw = reinterpret_cast<const RowRecWide&>(a);
    const RowRecWide& bw = reinterpret_cast<const RowRecWide&>(bq);
    is_section sec;
    sec.vT = vT;
    sec.type = type;
    sec.vB = vB;
    { /* ComputeMean, :47-60 */
        const float sd = a.S - bq.S;
        if (P.invalid >= 0) {
            const float valid_dif = a.V - bq.V;
            sec.disparity = (valid_dif == 0) ? 0 : sd / valid_dif;
        } else {
            sec.disparity = sd / (float)(vT + 1 - vB);
        }
    }
    sec.cost = __builtin_fminf(cost, 1e4f);
    const int hgt = vT + 1 - vB;
    const float meanx = wide ? (float)(aw.MX - bw.MX) : (a.MX - bq.MX);
    const float meany = wide ? (float)(aw.MY - bw.MY) : (a.MY - bq.MY);
    sec.instance_meanx = meanx / (float)hgt;
    sec.instance_meany = meany / (float)hgt;
    if (sec.type == IS_GROUND) { /* GetGroundSegmentationClass, Cityscapes.h:52-59 */
        const float cost_road = wide ? (float)(aw.Fg0 - bw.Fg0) : (a.Fg0 - bq.Fg0);
        const float cost_sidewalk = wide ? (float)(aw.Fg1 - bw.Fg1) : (a.Fg1 - bq.Fg1);
        sec.semantic_class = (cost_road < cost_sidewalk) ? 0 : 1;
    } else if (sec.type == IS_SKY || sec.disparity < 1.0f) { /* :894-902 */
        sec.type = IS_SKY;
        sec.semantic_class = 10;
    } else { /* GetObjectSegmentationClass, Cityscapes.h:85-111 */
        const float meanx2 = wide ? (float)(aw.MX2 - bw.MX2)
                                  : ((a.MX2h - bq.MX2h) + (a.MX2l - bq.MX2l));
        const float meany2 = wide ? (float)(aw.MY2 - bw.MY2)
                                  : ((a.MY2h - bq.MY2h) + (a.MY2l - bq.MY2l));
        const float height = (float)hgt;
        const float ic = P.iw * (meanx2 - meanx * meanx / height + meany2 - meany * meany / height);
        const float nic = P.iw * (float)(a.Fnic - bq.Fnic);
        float min_cost = IS_INF;
        int min_class = 2;
#pragma unroll
        for (int c = 0; c < IS_N_ON; c++) {
            float cs = 0.0f;
            cs += nic;
            cs += wide ? (float)(aw.Fon[c] - bw.Fon[c]) : (a.Fon[c] - bq.Fon[c]);
            if (min_cost > cs) { min_cost = cs; min_class = 2 + c; }
        }
#pragma unroll
        for (int c = 0; c < IS_N_OI; c++) {
            float cs = 0.0f;
            cs += ic;
            cs += wide ? (float)(aw.Foi[c] - bw.Foi[c]) : (a.Foi[c] - bq.Foi[c]);
            if (min_cost > cs) { min_cost = cs; min_class = 11 + c; }
        }
        sec.semantic_class = min_class;
    }
    return sec;
}

/* One wavefront per column.  The reference lets thread 0 do everything serially
 * (StixelsKernels.cu:843-955); only the index chase is inherently serial, so: the column's
 * tables are staged in LDS (coalesced), lane 0 walks the chain in LDS and records the cuts,
 * then the lanes build the Sections in parallel (one per lane). */
__global__ __launch_bounds__(64) void k_backtrace(const DevParams P, int ncols, int pairwise,
                                                  const RowRec* __restrict__ recs,
                                                  const float* __restrict__ cost_table,
                                                  const int32_t* __restrict__ index_table,
                                                  const int* __restrict__ col_flags,
                                                  is_section* __restrict__ sections) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int colg = blockIdx.x;
    if (colg >= ncols) return;
    const int lane = threadIdx.x;
    const int H = P.H, S = P.S;
    float* s_cost = (float*)smem;           /* [3H] */
    int* s_idx = (int*)(s_cost + 3 * H);    /* [3H] */
    int* s_cut = s_idx + 3 * H;             /* [S][3]: vT, vB, type */
    int* s_n = s_cut + 3 * S;               /* [1] */
    const bool wide = col_flags[colg] != 0; /* generic record encoding, see RowRec */
    const RowRec* rcol = recs + (size_t)colg * (H + 1);
    const float* ct = cost_table + (size_t)colg * H * 3;
    const int32_t* it = index_table + (size_t)colg * H * 3;
    is_section* out = sections + (size_t)colg * S;
    for (int i = lane; i < 3 * H; i += 64) {
        s_cost[i] = ct[i];
        s_idx[i] = it[i];
    }
    __syncthreads();
    if (lane == 0) {
        int vT = H - 1;
        const float last_ground = s_cost[vT * 3 + IS_GROUND];
        const float last_object = s_cost[vT * 3 + IS_OBJECT];
        const float last_sky = s_cost[vT * 3 + IS_SKY];
        int type = IS_OBJECT; /* :854-861 */
        if (last_ground < last_object) type = IS_GROUND;
        if (last_sky < __builtin_fminf(last_ground, last_object)) type = IS_SKY;
        int n = 0;
        int prev_vT;
        do {
            const int raw = s_idx[vT * 3 + type];
            int vB, prev_type;
            if (pairwise) {
                vB = raw / 3;
                prev_type = raw % 3;
            } else {
                /* unary: index_table holds the winning vB; the predecessor type is the arg-min
                 * of the FINAL cost_table[vB-1], tie rules of :723-727, 769-773, 828-835 */
                vB = raw;
                prev_type = IS_OBJECT;
                if (vB > 0) {
                    const float cG = s_cost[(vB - 1) * 3 + IS_GROUND];
                    const float cO = s_cost[(vB - 1) * 3 + IS_OBJECT];
                    if (cG < cO) prev_type = IS_GROUND;
                    if (type == IS_OBJECT) {
                        const float cS = s_cost[(vB - 1) * 3 + IS_SKY];
                        if (cS < __builtin_fminf(cG, cO)) prev_type = IS_SKY;
                    }
                }
            }
            s_cut[n * 3 + 0] = vT; s_cut[n * 3 + 1] = vB; s_cut[n * 3 + 2] = type;
            prev_vT = vB - 1;
            type = prev_type;
            vT = prev_vT;
            n++;
        } while (prev_vT != -1 && n < S - 1); /* the reference asserts i < max_sections (:950) */
        *s_n = n;
    }
    __syncthreads();
    const int n = *s_n;
    for (int i = lane; i <= n; i += 64) {
        is_section sec;
        if (i < n) {
            const int vT = s_cut[i * 3 + 0], vB = s_cut[i * 3 + 1], type = s_cut[i * 3 + 2];
            sec = make_section(P, rcol, wide, vT, vB, type, s_cost[vT * 3 + type]);
        } else { /* terminator, :952-954 */
            sec.type = -1; sec.vB = 0; sec.vT = 0; sec.disparity = 0.0f;
            sec.semantic_class = 0; sec.cost = 0.0f; sec.instance_meanx = 0.0f; sec.instance_meany = 0.0f;
        }
        out[i] = sec;
    }
}

/* ====================================================================================== */
/* Instance candidates in canonical (column, section) order, reference layout              */
/* (StixelsKernels.cu:926-942; one workgroup per image)                                    */
/* ====================================================================================== */
__global__ __launch_bounds__(256) void k_compact_instances(
    const DevParams P, const is_section* __restrict__ sections, float* __restrict__ com,
    int32_t* __restrict__ indices, uint8_t* __restrict__ core, int32_t* __restrict__ per_class) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* s_cnt = (int*)smem; /* [C][8] counts, then exclusive offsets */
    const int C = P.C, S = P.S;
    const is_section* sec = sections; /* already offset to the image by the host */
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        int cnt[IS_INSTANCE_CLASSES];
#pragma unroll
        for (int k = 0; k < IS_INSTANCE_CLASSES; k++) cnt[k] = 0;
        for (int i = 0; i < S; i++) {
            const is_section s = sec[(size_t)c * S + i];
            if (s.type == -1) break;
            if (s.type == IS_OBJECT && s.semantic_class >= IS_FIRST_INSTANCE_CLASS) {
                const int k = s.semantic_class - IS_FIRST_INSTANCE_CLASS;
#pragma unroll
                for (int kk = 0; kk < IS_INSTANCE_CLASSES; kk++)
                    if (kk == k) cnt[kk]++;
            }
        }
#pragma unroll
        for (int k = 0; k < IS_INSTANCE_CLASSES; k++) s_cnt[c * IS_INSTANCE_CLASSES + k] = cnt[k];
    }
    __syncthreads();
    if (threadIdx.x < IS_INSTANCE_CLASSES) {
        int run = 0;
        for (int c = 0; c < C; c++) {
            const int n = s_cnt[c * IS_INSTANCE_CLASSES + threadIdx.x];
            s_cnt[c * IS_INSTANCE_CLASSES + threadIdx.x] = run;
            run += n;
        }
        if (per_class) per_class[threadIdx.x] = run;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        int off[IS_INSTANCE_CLASSES];
#pragma unroll
        for (int k = 0; k < IS_INSTANCE_CLASSES; k++) off[k] = s_cnt[c * IS_INSTANCE_CLASSES + k];
        for (int i = 0; i < S; i++) {
            const is_section s = sec[(size_t)c * S + i];
            if (s.type == -1) break;
            if (s.type == IS_OBJECT && s.semantic_class >= IS_FIRST_INSTANCE_CLASS) {
                const int k = s.semantic_class - IS_FIRST_INSTANCE_CLASS;
                int idx = 0;
#pragma unroll
                for (int kk = 0; kk < IS_INSTANCE_CLASSES; kk++)
                    if (kk == k) idx = off[kk]++;
                const size_t slot = (size_t)k * C * S + idx;
                if (com) { com[slot * 2] = s.instance_meanx; com[slot * 2 + 1] = s.instance_meany; }
                if (indices) { indices[slot * 2] = c; indices[slot * 2 + 1] = i; }
                if (core) core[slot] = (s.vT + 1 - s.vB) >= P.size_filter;
            }
        }
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

/* ====================================================================================== */
/* launch helpers (called from is_core.hip)                                                */
/* ====================================================================================== */
extern "C" {

size_t isk_prepare_lds_bytes(const DevParams* P) {
    return sizeof(float) * (size_t)P->P2 * 3 + sizeof(int32_t) * (size_t)P->CH * P->P2S + 64;
}
size_t isk_unary_lds_bytes(const DevParams* P) {
    const size_t rcp = sizeof(float) * (((size_t)P->H + 1 + 3) & ~(size_t)3);
    const size_t tile = sizeof(float) * (size_t)IS_TILE * (P->D + 1);
    const size_t merge = (size_t)IS_UNARY_WAVES * 3 * 64 * 8; /* aliases the tile after the loop */
    return rcp + (tile > merge ? tile : merge) + 16;
}
size_t isk_pairwise_lds_bytes(const DevParams* P, int nwaves) { return isk_unary_lds_bytes(P); }
size_t isk_phase2_lds_bytes(const DevParams* P) {
    return sizeof(double) * 2 * IS_LOG_TABLE_SIZE + sizeof(float) * (2 * ((size_t)P->H + 1) + P->D) + 16;
}

hipError_t isk_launch_join(const float* big, float* joined, int H, int W, int C, int step,
                           int margin, int median, float invalid, int n_images,
                           hipStream_t stream) {
    dim3 grid((H + JOIN_ROWS - 1) / JOIN_ROWS, (C + JOIN_COLS - 1) / JOIN_COLS, n_images);
    hipLaunchKernelGGL(k_join_columns, grid, dim3(256), 0, stream, big, joined, H, W, C, step,
                       margin, median, invalid);
    return hipGetLastError();
}

hipError_t isk_launch_prepare(const DevParams* P, int ncols, const float* joined,
                              const int32_t* seg, const float* ground, const int* vhor,
                              const float* cost_T, RowRec* recs, float* lutT,
                              int* col_flags, float* sv_arr, hipStream_t stream, hipStream_t aux,
                              hipEvent_t ev_fork, hipEvent_t ev_join) {
    /* The two prepare kernels are independent.  With few columns (a single frame = 256) neither
     * fills the chip and both are latency chains, so they run side by side on two streams; with
     * many columns they are throughput-bound (HBM writes) and stay in order on one stream. */
    const bool side_by_side = aux != nullptr && ncols < IS_PREPARE_OVERLAP_MAX_COLS;
    hipError_t e;
    hipStream_t lut_stream = stream;
    if (side_by_side) {
        if ((e = hipEventRecord(ev_fork, stream)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(aux, ev_fork, 0)) != hipSuccess) return e;
        lut_stream = aux;
    }
    hipLaunchKernelGGL(k_object_lut, dim3(ncols, (P->D + 63) / 64), dim3(64), 0, lut_stream, *P,
                       joined, cost_T, lutT);
    hipLaunchKernelGGL(k_prepare_columns, dim3(ncols), dim3(PREP_THREADS),
                       isk_prepare_lds_bytes(P), stream, *P, joined, seg, ground, vhor, recs,
                       col_flags, sv_arr);
    if (side_by_side) {
        if ((e = hipEventRecord(ev_join, aux)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(stream, ev_join, 0)) != hipSuccess) return e;
    }
    return hipGetLastError();
}

hipError_t isk_launch_priors(const DevParams* P, const float* ground, PriorRec* priors,
                             int n_images, hipStream_t stream) {
    const int n = n_images * P->H;
    hipLaunchKernelGGL(k_prior_tables, dim3((n + 255) / 256), dim3(256), 0, stream, *P, ground,
                       priors, n_images);
    return hipGetLastError();
}

hipError_t isk_launch_dp_unary(const DevParams* P, int ncols, int nwaves, const RowRec* recs,
                               const float* lutT, const float* rcp, const int* vhor,
                               const int* col_flags, float* cost_table, int32_t* index_table,
                               hipStream_t stream) {
    const int groups = (ncols + 7) / 8;
    /* one tile pair (big, small) per workgroup: equal-length workgroups pack best; measured on
     * MI355X at batch 32: 1 pair 9.98 ms, 2 pairs 10.10 ms, 4 pairs 10.55 ms, single tiles 11.1 ms */
    const int npairs = (P->ntiles + 1) / 2;
    const int pairs_per_wg = 1;
    const int wg_per_col = (npairs + pairs_per_wg - 1) / pairs_per_wg;
    const dim3 grid(groups * 8 * wg_per_col);
    const size_t lds = isk_unary_lds_bytes(P);
    /* D <= 128: the vB-side lutT row travels in two registers per lane (LutRow<2>); wider
     * tables gather per lane */
#define IS_LAUNCH_UNARY(INV, NR)                                                                   \
    do {                                                                                           \
        hipLaunchKernelGGL((k_dp_unary<INV, NR, true>), grid, dim3(nwaves * 64), lds, stream, *P,  \
                           ncols, recs, lutT, rcp, vhor, col_flags, cost_table, index_table,       \
                           pairs_per_wg);                                                          \
        hipLaunchKernelGGL((k_dp_unary<INV, NR, false>), grid, dim3(nwaves * 64), lds, stream, *P, \
                           ncols, recs, lutT, rcp, vhor, col_flags, cost_table, index_table,       \
                           pairs_per_wg);                                                          \
    } while (0)
    if (P->D <= 128) {
        if (P->invalid >= 0) IS_LAUNCH_UNARY(true, 2); else IS_LAUNCH_UNARY(false, 2);
    } else {
        if (P->invalid >= 0) IS_LAUNCH_UNARY(true, 0); else IS_LAUNCH_UNARY(false, 0);
    }
#undef IS_LAUNCH_UNARY
    return hipGetLastError();
}

hipError_t isk_launch_dp_pairwise(const DevParams* P, int ncols, int nwaves, const RowRec* recs,
                                  const float* lutT, const PriorRec* priors, const float* odr,
                                  const float* rcp, const float* sv_arr, const int* vhor,
                                  const int* col_flags, StepRec* steps, float* part_cost,
                                  int* part_idx, float* cost_table, int32_t* index_table,
                                  hipStream_t stream, hipStream_t aux, hipEvent_t ev_fork,
                                  hipEvent_t ev_join) {
    const size_t lds1 = isk_pairwise_lds_bytes(P, nwaves);
    const size_t lds2 = isk_phase2_lds_bytes(P);
    /* Columns are independent: with enough of them the batch is cut in two halves whose
     * phase-1 / phase-2 chains run on two streams, the second one phase behind the first, so
     * that the issue-bound phase 1 of one half shares the CUs with the latency-bound serial
     * phase 2 of the other. */
    /* few columns: nsplit workgroups per (column, tile) in phase 1, up to ~one workgroup per CU x4 */
    int nsplit = IS_PW_SPLIT_TARGET_WGS / (ncols > 0 ? ncols : 1);
    nsplit = nsplit < 1 ? 1 : (nsplit > IS_PW_MAX_SPLIT ? IS_PW_MAX_SPLIT : nsplit);
    const bool split = aux != nullptr && ncols >= 2 * IS_PAIRWISE_SPLIT_MIN_COLS;
    const int c_mid = split ? (ncols / 2) : ncols;
    hipError_t e;
/* phase 1 with the vB-side lutT row in registers (LutRow<2>) measured SLOWER than the per-lane
 * gather on MI355X (41.4 vs 38.0 ms per 64 frames): the pick costs more VALU than the gather's
 * address arithmetic and phase 1 is issue-bound; kept selectable for later rounds */
#define IS_PW_PHASE1_ROW_REGS 0
#define IS_LAUNCH_P1(INV, c0, c1, st)                                                              \
    do {                                                                                           \
        if (IS_PW_PHASE1_ROW_REGS && P->D <= 128)                                                  \
            hipLaunchKernelGGL((k_pw_phase1<INV, 2>), dim3(((c1) - (c0)) * nsplit),                \
                               dim3(nwaves * 64), lds1, st, *P, c0, c1, tile, nsplit, recs, lutT,  \
                               steps, rcp, vhor, col_flags, part_cost, part_idx);                  \
        else                                                                                       \
            hipLaunchKernelGGL((k_pw_phase1<INV, 0>), dim3(((c1) - (c0)) * nsplit),                \
                               dim3(nwaves * 64), lds1, st, *P, c0, c1, tile, nsplit, recs, lutT,  \
                               steps, rcp, vhor, col_flags, part_cost, part_idx);                  \
    } while (0)
#define IS_LAUNCH_P2(INV, c0, c1, st)                                                              \
    hipLaunchKernelGGL(k_pw_phase2<INV>, dim3((c1) - (c0)), dim3(64), lds2, st, *P, c0, c1, tile,  \
                       nsplit, recs, lutT, priors, odr, rcp, sv_arr, vhor, col_flags, part_cost,   \
                       part_idx, steps, cost_table, index_table)
    const bool inv = P->invalid >= 0;
    for (int tile = 0; tile < P->ntiles; tile++) {
        if (inv) IS_LAUNCH_P1(true, 0, c_mid, stream); else IS_LAUNCH_P1(false, 0, c_mid, stream);
        if (split && tile == 0) { /* the second half starts one phase behind the first */
            if ((e = hipEventRecord(ev_fork, stream)) != hipSuccess) return e;
            if ((e = hipStreamWaitEvent(aux, ev_fork, 0)) != hipSuccess) return e;
        }
        if (inv) IS_LAUNCH_P2(true, 0, c_mid, stream); else IS_LAUNCH_P2(false, 0, c_mid, stream);
        if (split) {
            if (inv) IS_LAUNCH_P1(true, c_mid, ncols, aux); else IS_LAUNCH_P1(false, c_mid, ncols, aux);
            if (inv) IS_LAUNCH_P2(true, c_mid, ncols, aux); else IS_LAUNCH_P2(false, c_mid, ncols, aux);
        }
    }
#undef IS_LAUNCH_P1
#undef IS_LAUNCH_P2
    if (split) {
        if ((e = hipEventRecord(ev_join, aux)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(stream, ev_join, 0)) != hipSuccess) return e;
    }
    return hipGetLastError();
}

hipError_t isk_launch_backtrace(const DevParams* P, int ncols, int pairwise, const RowRec* recs,
                                const float* cost_table, const int32_t* index_table,
                                const int* col_flags, is_section* sections, hipStream_t stream) {
    const size_t lds = sizeof(int) * (6 * (size_t)P->H + 3 * (size_t)P->S + 4);
    hipLaunchKernelGGL(k_backtrace, dim3(ncols), dim3(64), lds, stream, *P, ncols, pairwise, recs,
                       cost_table, index_table, col_flags, sections);
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

hipError_t isk_launch_compact(const DevParams* P, const is_section* sections_img, float* com,
                              int32_t* indices, uint8_t* core, int32_t* per_class,
                              hipStream_t stream) {
    const size_t lds = sizeof(int) * (size_t)P->C * IS_INSTANCE_CLASSES + 16;
    hipLaunchKernelGGL(k_compact_instances, dim3(1), dim3(256), lds, stream, *P, sections_img, com,
                       indices, core, per_class);
    return hipGetLastError();
}

int isk_debug_occupancy(const DevParams* P, int nwaves) {
    int nb = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_dp_unary<false, 2, true>,
                                                 nwaves * 64, isk_unary_lds_bytes(P));
    return nb;
}

hipError_t isk_set_lds_limits(const DevParams* P, int nwaves_pair) {
    hipError_t e;
    const int a = (int)isk_prepare_lds_bytes(P);
    e = hipFuncSetAttribute((const void*)k_prepare_columns, hipFuncAttributeMaxDynamicSharedMemorySize, a);
    if (e != hipSuccess) return e;
    const int b = (int)isk_unary_lds_bytes(P);
#define IS_SET_UNARY_LDS(INV, NR, FC)                                                             \
    e = hipFuncSetAttribute((const void*)k_dp_unary<INV, NR, FC>,                                 \
                            hipFuncAttributeMaxDynamicSharedMemorySize, b);                       \
    if (e != hipSuccess) return e
    IS_SET_UNARY_LDS(true, 2, true); IS_SET_UNARY_LDS(true, 2, false);
    IS_SET_UNARY_LDS(false, 2, true); IS_SET_UNARY_LDS(false, 2, false);
    IS_SET_UNARY_LDS(true, 0, true); IS_SET_UNARY_LDS(true, 0, false);
    IS_SET_UNARY_LDS(false, 0, true); IS_SET_UNARY_LDS(false, 0, false);
#undef IS_SET_UNARY_LDS
    e = hipFuncSetAttribute((const void*)k_backtrace, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(sizeof(int) * (6 * (size_t)P->H + 3 * (size_t)P->S + 4)));
    if (e != hipSuccess) return e;
    const int c = (int)isk_pairwise_lds_bytes(P, nwaves_pair);
    e = hipFuncSetAttribute((const void*)k_pw_phase1<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, c);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k_pw_phase1<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, c);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k_pw_phase1<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, c);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k_pw_phase1<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, c);
    return e;
}

} /* extern "C" */
