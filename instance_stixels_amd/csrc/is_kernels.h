/*
 * is_kernels.h -- shared device code of the gfx950 (CDNA4, wave64) kernels of the
 * Instance-Stixels column DP (is_k_*.hip).  Nothing here is translated from the reference's
 * CUDA: the reference runs one 1024-thread block per column with a barrier per vB and 126 global
 * loads per (vB, vT) pair (/root/reference/InstanceStixels/src/StixelsKernels.cu:600-839); the
 * kernels restructure the same arithmetic (bit-exactly, see DESIGN.md "Exact rewrites") as
 *
 *   is_k_frontend.hip   k_join_columns      A3   StixelsKernels.cu:980-1095   LDS-transposed
 *                       k_flip_and_pad, k_vdisp_*  
 *   is_k_prepare.hip    k_prepare_columns   A4-A6 StixelsKernels.cu:371-469 and
 *                                           StixelsKernels.h:73-103: per-row boundary records
 *                       k_object_lut        A4   StixelsKernels.cu:236-296, 959-978
 *                       k_prior_tables      A9   StixelsKernels.cu:88-199 (DP-state independent)
 *   is_k_unary.hip      k_dp_unary          A7-A9 StixelsKernels.cu:477-839, PAIRWISE=false:
 *                                           independent (column, tile pair) work items, one lane
 *                                           per vT, vB-side operands in SGPRs via scalar loads,
 *                                           vT-side LUT rows in LDS
 *   is_k_pairwise.hip   k_pw_phase1/2       A7-A9 PAIRWISE=true: per 64-row tile, a parallel
 *                                           launch for segments starting in earlier tiles + a
 *                                           one-wave-per-column diagonal walk
 *   is_k_backtrace.hip  k_backtrace         A10  StixelsKernels.cu:843-955
 *                       k_compact_instances A10  StixelsKernels.cu:926-942, canonical order (R9)
 *
 * This header: the segment evaluation and the LUT / record access helpers both DP kernels use.
 * Numerics contract: IEEE fp32, no contraction (-ffp-contract=off), correctly rounded
 * division, no fast-math; integer sums in wrapping int32 / int64 like the reference.
 */
#ifndef IS_KERNELS_H_
#define IS_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "instance_stixels_core.h"
#include "is_device.h"
#include "is_numerics.h"

#define IS_INF (__builtin_inff())

typedef const __attribute__((address_space(4))) RowRec* crec_t;     /* scalar-load view */
typedef const __attribute__((address_space(4))) PriorRec* cprior_t;

/* workgroup LDS sizes (defined next to the kernels that use them) */
extern "C" {
size_t isk_prepare_lds_bytes(const DevParams* P);
size_t isk_unary_lds_bytes(const DevParams* P);
size_t isk_pairwise_lds_bytes(const DevParams* P, int nwaves);
size_t isk_phase2_lds_bytes(const DevParams* P);
size_t isk_unary_fast_lds_bytes(const DevParams* P, int chunk_rows);
int isk_unary_fast_chunk_rows(const DevParams* P);
hipError_t isk_set_lds_unary_fast(const DevParams* P);
}

/* ====================================================================================== */
/* Segment evaluation shared by both DP kernels                                            */
/* ====================================================================================== */
struct SegTerms {
    float seg_g, seg_o, seg_s; /* semantic + instance terms of the three geometric classes */
    float gd, sd;              /* ground / sky data terms                                   */
    float mean;                /* un-floored, clamped (>= 0) object mean disparity          */
    int fni;                   /* floor(mean), clamped to [0, D-1]                          */
    /* float(min_c DownsampledSum_c) of the class groups: non-decreasing when the segment grows
     * in a FAST column (class values >= 0) -- the lower bounds of the branch-and-bound */
    float f_g, f_on, f_oi, f_sky;
    /* also non-decreasing when the segment grows (lemmas L2 / L4): on = nic +
     * f_on exactly (nic: iw * float(exact non-negative integer sum)), the instance term ic up to
     * its rounding error (a sum of squared deviations only grows) */
    float on, ic;
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

/* ComputeMean with an invalid-disparity value (StixelsKernels.cu:47-60) in a FAST column: the divisor is the
 * number of valid rows of the segment -- an exact integer in [0, h] (the prefix of 0 / 1 values is exact in
 * fp32) -- so the exact-division shortcut applies with r = RN(1 / valid) from the 1/h table `rcp_tab`
 * (entries 0 .. at least the segment's height): one table read, one multiplication and two FMAs instead of
 * the IEEE division sequence (v_div_scale x 2, v_rcp, four FMAs, v_div_fmas, v_div_fixup), bit for bit the
 * same quotient (tools/verify_exact_division.c covers every divisor <= 11000, not only the height).  A
 * segment without a valid row has mean 0 (:55); the table entry of 0 is then read but never used. */
__device__ __forceinline__ float mean_valid_fast(float sdif, float valid_dif, const float* rcp_tab) {
    const float rv = rcp_tab[cvt_u32_sat(valid_dif)];
    const float q = fast_div(sdif, valid_dif, rv);
    return (valid_dif == 0) ? 0.0f : q;
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
/* rcp_tab (HAS_INVALID, FAST columns): the 1/h table for mean_valid_fast, or null = IEEE division */
template <bool FAST, bool HAS_INVALID>
__device__ __forceinline__ SegTerms eval_segment(const RowRec& my, const RowRec& rb, float height,
                                                 float r, int D, float iw, const float* rcp_tab = nullptr) {
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

    t.f_g = f_g; t.f_on = f_on; t.f_oi = f_oi; t.f_sky = f_sky;
    t.seg_g = f_g + nic;
    const float on = nic + f_on;
    const float oi = ic + f_oi;
    t.on = on; t.ic = ic;
    t.seg_o = FAST ? __builtin_fminf(oi, on) : ((oi < on) ? oi : on); /* both finite when FAST */
    t.seg_s = f_sky + nic;

    t.gd = my.G - rb.G;
    t.sd = my.K - rb.K;
    float mean; /* ComputeMean, :47-60 */
    if (HAS_INVALID) {
        const float valid_dif = my.V - rb.V;
        if (FAST && rcp_tab != nullptr) mean = mean_valid_fast(my.S - rb.S, valid_dif, rcp_tab);
        else mean = (valid_dif == 0) ? 0 : (my.S - rb.S) / valid_dif;
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
        /* four sweeps per round trip: all four loads are issued before the first LDS store (a
         * load -> wait -> store loop costs one full memory latency per sweep, and its vmcnt(0)
         * also waits for every prefetch the caller has in flight) */
        for (int rb = r0; rb < IS_TILE; rb += 4 * dr) {
            float4 x[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int r = rb + k * dr;
                const int v = min(tile_lo + 1 + min(r, IS_TILE - 1), H);
                x[k] = *reinterpret_cast<const float4*>(lcol + (size_t)v * D + f);
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int r = rb + k * dr;
                if (r < IS_TILE) {
                    float* d = s_tile + (r) * DP + f;
                    d[0] = x[k].x; d[1] = x[k].y; d[2] = x[k].z; d[3] = x[k].w;
                }
            }
        }
    } else if ((nthreads % D) == 0) {
        const int r0 = tid / D, f = tid - r0 * D;
        const int dr = nthreads / D;
        for (int rb = r0; rb < IS_TILE; rb += 8 * dr) {
            float x[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int r = rb + k * dr;
                x[k] = lcol[(size_t)min(tile_lo + 1 + min(r, IS_TILE - 1), H) * D + f];
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int r = rb + k * dr;
                if (r < IS_TILE) s_tile[(r) * DP + f] = x[k];
            }
        }
    } else {
        for (int i = tid; i < IS_TILE * D; i += nthreads) {
            const int r = i / D, f = i - r * D;
            const int v = min(tile_lo + 1 + r, H);
            s_tile[(r) * DP + f] = lcol[(size_t)v * D + f];
        }
    }
}

/* s_rcp[0..H] <- rcp[0..H], four elements per thread and round trip (see stage_lut_tile) */
__device__ __forceinline__ void stage_rcp(float* s_rcp, const float* __restrict__ rcp, int H, int tid,
                                          int nthreads) {
    for (int ib = tid; ib <= H; ib += 4 * nthreads) {
        float x[4];
#pragma unroll
        for (int k = 0; k < 4; k++) x[k] = rcp[min(ib + k * nthreads, H)];
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (ib + k * nthreads <= H) s_rcp[ib + k * nthreads] = x[k];
    }
}

#ifndef IS_STAGE_SPREAD
#define IS_STAGE_SPREAD 1
#endif
/* The tile and the 1/h table behind ONE memory round trip: staged one after the other, the
 * second one's loads are only issued after the first one's vmcnt(0) wait (measured in the unary
 * ring kernel: 19 % + 15 % of a workgroup's life for the two).  Common shapes (16-byte path of
 * stage_lut_tile in one batch, H + 1 <= 4 * nthreads) take the fused path. */
__device__ __forceinline__ void stage_tile_and_rcp(float* s_tile, float* s_rcp,
                                                   const float* __restrict__ lcol,
                                                   const float* __restrict__ rcp, int tile_lo, int H,
                                                   int D, int tid, int nthreads) {
    const int quads = D >> 2;
    const bool fused = (D & 3) == 0 && ((4 * nthreads) % D) == 0 && (nthreads / quads) * 4 >= IS_TILE &&
                       H + 1 <= 4 * nthreads;
    if (!fused) {
        stage_rcp(s_rcp, rcp, H, tid, nthreads);
        stage_lut_tile<true>(s_tile, lcol, tile_lo, H, D, tid, nthreads);
        return;
    }
    const int DP = D + 1;
    float rc[4];
    float4 x[4];
#pragma unroll
    for (int k = 0; k < 4; k++) rc[k] = rcp[min(tid + k * nthreads, H)];
#if IS_STAGE_SPREAD
    /* D = 128, 512 threads: a group of 8 lanes takes one 128-byte line, a wave 8 ROWS x 8 chunks, its four
     * loads the four quarters of those rows: the dword stores below then hit banks (row + 4 chunk + j)
     * mod 32 -- 8 rows x 8 chunks spread over all of them -- instead of one row's 32 chunks at a stride
     * of four dwords (4-way conflicts on every staging store) */
    const bool spread = quads == 32 && nthreads == 8 * IS_TILE;
    const int sr = (tid >> 6) * 8 + ((tid & 63) >> 3), sq = tid & 7;
#else
    const bool spread = false;
    const int sr = 0, sq = 0;
#endif
    const int r0 = tid / quads, f0 = (tid - r0 * quads) * 4;
    const int dr = nthreads / quads;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int r = spread ? sr : min(r0 + k * dr, IS_TILE - 1);
        const int f = spread ? (sq + 8 * k) * 4 : f0;
        x[k] = *reinterpret_cast<const float4*>(lcol + (size_t)min(tile_lo + 1 + r, H) * D + f);
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (tid + k * nthreads <= H) s_rcp[tid + k * nthreads] = rc[k];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int r = spread ? sr : r0 + k * dr;
        const int f = spread ? (sq + 8 * k) * 4 : f0;
        if (r < IS_TILE) {
            float* d = s_tile + (r) * DP + f;
            d[0] = x[k].x; d[1] = x[k].y; d[2] = x[k].z; d[3] = x[k].w;
        }
    }
}

/* The fn window `win` (two halves of IS_P1_WIN / 2 columns, is_device.h) of lutT rows tile_lo + 1 .. tile_lo + 64 (row
 * stride IS_P1_WIN + 1) and the 1/h table behind one memory round trip.  The halves start at multiples of 4: eight
 * lanes fetch the 128 bytes of one half with 16-byte loads, sixteen a row's window, a wave four rows; the dword
 * stores of a wave then hit banks (row + 4 q + j) mod 64, q = 0 .. 15, four consecutive rows: all different. */
__device__ __forceinline__ void stage_window_and_rcp(float* s_tile, float* s_rcp, const float* __restrict__ lcol,
                                                     const float* __restrict__ rcp, int tile_lo, int H, int D,
                                                     int win, int tid, int nthreads) {
    constexpr int WQ = IS_P1_WIN / 4; /* 16-byte chunks per row */
    constexpr int WPs = IS_P1_WIN + 1;
    constexpr int NX = 4;             /* chunks per thread and batch */
    constexpr int NR = 5;             /* 1/h entries per thread: 1025 <= 5 * 256 */
    const bool one_trip = H + 1 <= NR * nthreads;
    float rc[NR];
    if (one_trip) {
#pragma unroll
        for (int k = 0; k < NR; k++) rc[k] = rcp[min(tid + k * nthreads, H)];
    }
    const int q = tid & (WQ - 1), r0 = tid / WQ, dr = nthreads / WQ; /* (nthreads: a multiple of 64) */
    for (int rb = r0; rb < IS_TILE; rb += NX * dr) {
        float4 x[NX];
#pragma unroll
        for (int k = 0; k < NX; k++) {
            const int r = min(rb + k * dr, IS_TILE - 1);
            x[k] = *reinterpret_cast<const float4*>(lcol + (size_t)min(tile_lo + 1 + r, H) * D + IS_WIN_COL(win, 4 * q));
        }
#pragma unroll
        for (int k = 0; k < NX; k++) {
            const int r = rb + k * dr;
            if (r < IS_TILE) {
                float* d = s_tile + (r) * WPs + 4 * q;
                d[0] = x[k].x; d[1] = x[k].y; d[2] = x[k].z; d[3] = x[k].w;
            }
        }
    }
    if (one_trip) {
#pragma unroll
        for (int k = 0; k < NR; k++)
            if (tid + k * nthreads <= H) s_rcp[tid + k * nthreads] = rc[k];
    } else {
        stage_rcp(s_rcp, rcp, H, tid, nthreads);
    }
}

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

/* (best, best_v) <- (cost, vB) in the lanes with cost <= best: the update of a DESCENDING walk
 * over vB (among equal costs the smallest vB must win, as in the reference's ascending loop with
 * a strict <).  A +inf candidate "wins" against the initial +inf; the merge restores the initial
 * index for rows whose final cost is +inf. */
__device__ __forceinline__ void take_if_le(float& best, int& best_v, float cost, int vB) {
    unsigned long long saved;
    asm("s_mov_b64 %[sv], exec\n\t"
        "v_cmpx_le_f32_e32 %[c], %[b]\n\t"
        "v_mov_b32_e32 %[b], %[c]\n\t"
        "v_mov_b32_e32 %[i], %[vb]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [b] "+v"(best), [i] "+v"(best_v), [sv] "=&s"(saved)
        : [c] "v"(cost), [vb] "s"(vB)
        : "vcc");
}

__device__ __forceinline__ void take_if_le_v(float& best, int& best_v, float cost, int v) {
    unsigned long long saved;
    asm("s_mov_b64 %[sv], exec\n\t"
        "v_cmpx_le_f32_e32 %[c], %[b]\n\t"
        "v_mov_b32_e32 %[b], %[c]\n\t"
        "v_mov_b32_e32 %[i], %[v]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [b] "+v"(best), [i] "+v"(best_v), [sv] "=&s"(saved)
        : [c] "v"(cost), [v] "v"(v)
        : "vcc");
}

typedef const __attribute__((address_space(4))) PruneRec* cprune_t;

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

/* ====================================================================================== */
/* The vB record as DPP operands (pairwise phase 1, unary ring kernel)                      */
/* ====================================================================================== */
/* Lane l of every 16-lane row holds dwords (l & 15) [R0] and 16 + (l & 15) [R1] of the record of
 * vB; a subtraction takes dword k as `row_newbcast:k`.  See is_k_pairwise.hip for the why. */
template <int K>
__device__ __forceinline__ float dpp_sub(float mine, float R) {
    float d;
    asm volatile("v_subrev_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "=v"(d) : "v"(R), "v"(mine), "n"(K));
    return d;
}
/* The first DPP read of a record register in a step: gfx9 needs two wait states between a VALU
 * write of a VGPR (the register rotation's v_mov) and a DPP read of it; the compiler's hazard
 * recogniser does not look inside inline assembly, so the s_nop travels with the instruction. */
template <int K>
__device__ __forceinline__ float dpp_sub_first(float mine, float R) {
    float d;
    asm volatile("s_nop 1\n\tv_subrev_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "=v"(d) : "v"(R), "v"(mine), "n"(K));
    return d;
}
/* integer dword (Fnic): fetched with v_mov_b32_dpp and subtracted by an ordinary instruction --
 * v_subrev_u32_dpp returns a wrong operand on gfx950 (tools/ubench/dpp_sub_check.hip) */
template <int K>
__device__ __forceinline__ int dpp_sub_i_first(int mine, float R) {
    int d;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf"
                 : "=v"(d) : "v"(R), "n"(K));
    return mine - d;
}
/* v_min_f32 / v_min3_f32 as they are.  `__builtin_fminf` on a value that comes out of inline
 * assembly (the DPP subtractions) makes the compiler canonicalise both operands first (two extra
 * 4-cycle v_max_f32 x, x per minimum) because it cannot see that they are no signalling NaNs; the
 * operands here are differences / sums of computed fp32 values, whose NaNs (if any) are quiet, so
 * the bare instruction already has fminf's semantics (the other operand wins over a quiet NaN;
 * kernels run in IEEE mode).  Not volatile: unused results disappear. */
__device__ __forceinline__ float min_raw(float a, float b) {
    float d;
    asm("v_min_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ float min3_raw(float a, float b, float c) {
    float d;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

/* Lower bound of the object semantic term seg_o(vB', vT) for every vB' <= vB from the terms of
 * (vB, vT): min(on, fl(fl(ic - 3 E2) + f_oi)), E2 >= |computed ic - real ic| (PruneRec).  3 E2:
 * 2 E2 for the two rounding errors, one more for the rounding of the subtraction itself (an ulp
 * of ic is below E2 / 4). */
__device__ __forceinline__ float seg_o_lower_bound(const SegTerms& t, float E2x3) {
    return min_raw(t.on, (t.ic - E2x3) + t.f_oi);
}


/* what a step needs besides the object terms: the ground candidate's, the sky candidate's, both
 * (the first segment of the pairwise model, generic callers) or neither (tiles above the horizon) */
#define IS_WANT_GROUND 1
#define IS_WANT_SKY 2

/* eval_segment<true, HAS_INVALID> (is_kernels.h) with the vB record in (R0, R1): identical
 * operations in identical order, only the source of the vB operand differs.  WANT: which of the
 * ground / sky terms are computed at all -- the DPP instructions are `asm volatile` and would
 * otherwise be executed even where their result is never read. */
template <bool HAS_INVALID, int WANT = IS_WANT_GROUND | IS_WANT_SKY>
__device__ __forceinline__ SegTerms eval_segment_dpp(const RowRec& my, float R0, float R1, float height,
                                                     float r, int D, float iw, const float* rcp_tab = nullptr) {
    SegTerms t;
    const float nic = iw * (float)dpp_sub_i_first<3>(my.Fnic, R1);
    float f_g = 0.0f;
    float f_on;
    if (WANT & IS_WANT_GROUND) {
        const float d_g0 = dpp_sub_first<0>(my.Fg0, R0); /* (explicit statements: asm order = source order) */
        const float d_g1 = dpp_sub<1>(my.Fg1, R0);
        f_g = min_raw(d_g0, d_g1);
        f_on = dpp_sub<2>(my.Fon[0], R0);
    } else {
        f_on = dpp_sub_first<2>(my.Fon[0], R0);
    }
    {
        const float a1 = dpp_sub<3>(my.Fon[1], R0), a2 = dpp_sub<4>(my.Fon[2], R0);
        f_on = min3_raw(f_on, a1, a2);
        const float a3 = dpp_sub<5>(my.Fon[3], R0), a4 = dpp_sub<6>(my.Fon[4], R0);
        f_on = min3_raw(f_on, a3, a4);
        const float a5 = dpp_sub<7>(my.Fon[5], R0), a6 = dpp_sub<8>(my.Fon[6], R0);
        f_on = min3_raw(f_on, a5, a6);
        f_on = min_raw(f_on, dpp_sub<9>(my.Fon[7], R0));
    }
    float f_oi = dpp_sub<10>(my.Foi[0], R0);
    {
        const float a1 = dpp_sub<11>(my.Foi[1], R0), a2 = dpp_sub<12>(my.Foi[2], R0);
        f_oi = min3_raw(f_oi, a1, a2);
        const float a3 = dpp_sub<13>(my.Foi[3], R0), a4 = dpp_sub<14>(my.Foi[4], R0);
        f_oi = min3_raw(f_oi, a3, a4);
        const float a5 = dpp_sub<15>(my.Foi[5], R0), a6 = dpp_sub<0>(my.Foi[6], R1);
        f_oi = min3_raw(f_oi, a5, a6);
        f_oi = min_raw(f_oi, dpp_sub<1>(my.Foi[7], R1));
    }
    float f_sky = 0.0f;
    if (WANT & IS_WANT_SKY) f_sky = dpp_sub<2>(my.Fsky, R1);
    const float meanx = dpp_sub<8>(my.MX, R1);
    const float meany = dpp_sub<9>(my.MY, R1);
    const float d_x2h = dpp_sub<10>(my.MX2h, R1);
    const float d_x2l = dpp_sub<11>(my.MX2l, R1);
    const float d_y2h = dpp_sub<12>(my.MY2h, R1);
    const float d_y2l = dpp_sub<13>(my.MY2l, R1);
    const float meanx2 = d_x2h + d_x2l;
    const float meany2 = d_y2h + d_y2l;
    const float ic = iw * (meanx2 - fast_div(meanx * meanx, height, r) + meany2 -
                           fast_div(meany * meany, height, r));
    t.f_g = f_g; t.f_on = f_on; t.f_oi = f_oi; t.f_sky = f_sky;
    t.seg_g = f_g + nic;
    const float on = nic + f_on;
    const float oi = ic + f_oi;
    t.on = on; t.ic = ic;
    t.seg_o = min_raw(oi, on); /* both finite in a FAST column */
    t.seg_s = f_sky + nic;
    t.gd = 0.0f;
    t.sd = 0.0f;
    if (WANT & IS_WANT_GROUND) t.gd = dpp_sub<4>(my.G, R1);
    if (WANT & IS_WANT_SKY) t.sd = dpp_sub<5>(my.K, R1);
    float mean;
    if (HAS_INVALID) {
        const float valid_dif = dpp_sub<7>(my.V, R1);
        const float sdif = dpp_sub<6>(my.S, R1);
        if (rcp_tab != nullptr) mean = mean_valid_fast(sdif, valid_dif, rcp_tab);
        else mean = (valid_dif == 0) ? 0 : sdif / valid_dif;
    } else {
        mean = fast_div(dpp_sub<6>(my.S, R1), height, r);
    }
    t.fni = (int)min(cvt_u32_sat(mean), (unsigned)(D - 1));
    t.mean = __builtin_fmaxf(mean, 0.0f);
    return t;
}

/* eval_segment_dpp with the FIRST half of the vB record (the 16 class prefixes Fg0, Fg1, Fon[8],
 * Foi[0..5]) as SCALAR operands S: a subtraction with an SGPR operand issues in 2 cycles, with a
 * DPP operand in 4, and a step of the unary ring kernel is bound by exactly that.  Identical
 * operations in identical order per value.  The caller brings S in with srec_request() one step
 * ahead and srec_arrived() before the step (one s_load_dwordx16 from the record's line in global
 * memory -- the ring's LDS-DMA has brought that line into the L2 steps before). */
typedef float isk_f16v __attribute__((ext_vector_type(16)));
#ifndef IS_MIX_PK
#define IS_MIX_PK 1 /* the sixteen scalar-operand subtractions as eight v_pk_add_f32 (neg modifiers): -1.5 % */
#endif
__device__ __forceinline__ void srec_request(isk_f16v& S, const RowRec* grec) {
    asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(S) : "s"(grec));
}
/* the same for a value that is live: the old contents are consumed here, so the compiler keeps
 * every use of them in front of the request */
__device__ __forceinline__ void srec_request_next(isk_f16v& S, const RowRec* grec) {
    asm volatile("s_load_dwordx16 %0, %1, 0x0" : "+s"(S) : "s"(grec));
}
__device__ __forceinline__ void srec_arrived(isk_f16v& S) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(S)); }

template <bool HAS_INVALID, int WANT = IS_WANT_GROUND | IS_WANT_SKY>
__device__ __forceinline__ SegTerms eval_segment_mix(const RowRec& my, const isk_f16v& S, float R1,
                                                     float height, float r, int D, float iw,
                                                     const float* rcp_tab = nullptr) {
    SegTerms t;
    const float nic = iw * (float)dpp_sub_i_first<3>(my.Fnic, R1);
    float f_g = 0.0f;
#if IS_MIX_PK
    typedef float f2 __attribute__((ext_vector_type(2)));
#define IS_PK_SUB(a0, a1, k) (f2{a0, a1} - f2{S[k], S[(k) + 1]})
    if (WANT & IS_WANT_GROUND) {
        const f2 d = IS_PK_SUB(my.Fg0, my.Fg1, 0);
        f_g = __builtin_fminf(d.x, d.y);
    }
    float f_on, f_oi;
    {
        const f2 d0 = IS_PK_SUB(my.Fon[0], my.Fon[1], 2), d1 = IS_PK_SUB(my.Fon[2], my.Fon[3], 4),
                 d2 = IS_PK_SUB(my.Fon[4], my.Fon[5], 6), d3 = IS_PK_SUB(my.Fon[6], my.Fon[7], 8);
        f_on = __builtin_fminf(d0.x, d0.y);
        f_on = __builtin_fminf(__builtin_fminf(f_on, d1.x), d1.y);
        f_on = __builtin_fminf(__builtin_fminf(f_on, d2.x), d2.y);
        f_on = __builtin_fminf(__builtin_fminf(f_on, d3.x), d3.y);
        const f2 e0 = IS_PK_SUB(my.Foi[0], my.Foi[1], 10), e1 = IS_PK_SUB(my.Foi[2], my.Foi[3], 12),
                 e2 = IS_PK_SUB(my.Foi[4], my.Foi[5], 14);
        f_oi = __builtin_fminf(e0.x, e0.y);
        f_oi = __builtin_fminf(__builtin_fminf(f_oi, e1.x), e1.y);
        f_oi = __builtin_fminf(__builtin_fminf(f_oi, e2.x), e2.y);
    }
#undef IS_PK_SUB
#else
    if (WANT & IS_WANT_GROUND) f_g = __builtin_fminf(my.Fg0 - S[0], my.Fg1 - S[1]);
    float f_on = my.Fon[0] - S[2];
#pragma unroll
    for (int c = 1; c < IS_N_ON; c++) f_on = __builtin_fminf(f_on, my.Fon[c] - S[2 + c]);
    float f_oi = my.Foi[0] - S[10];
#pragma unroll
    for (int c = 1; c < 6; c++) f_oi = __builtin_fminf(f_oi, my.Foi[c] - S[10 + c]);
#endif
    {
        const float a6 = dpp_sub<0>(my.Foi[6], R1), a7 = dpp_sub<1>(my.Foi[7], R1);
        f_oi = min3_raw(f_oi, a6, a7);
    }
    float f_sky = 0.0f;
    if (WANT & IS_WANT_SKY) f_sky = dpp_sub<2>(my.Fsky, R1);
    const float meanx = dpp_sub<8>(my.MX, R1);
    const float meany = dpp_sub<9>(my.MY, R1);
    const float d_x2h = dpp_sub<10>(my.MX2h, R1);
    const float d_x2l = dpp_sub<11>(my.MX2l, R1);
    const float d_y2h = dpp_sub<12>(my.MY2h, R1);
    const float d_y2l = dpp_sub<13>(my.MY2l, R1);
    const float meanx2 = d_x2h + d_x2l;
    const float meany2 = d_y2h + d_y2l;
    const float ic = iw * (meanx2 - fast_div(meanx * meanx, height, r) + meany2 -
                           fast_div(meany * meany, height, r));
    t.f_g = f_g; t.f_on = f_on; t.f_oi = f_oi; t.f_sky = f_sky;
    t.seg_g = f_g + nic;
    const float on = nic + f_on;
    const float oi = ic + f_oi;
    t.on = on; t.ic = ic;
    t.seg_o = min_raw(oi, on); /* both finite in a FAST column */
    t.seg_s = f_sky + nic;
    t.gd = 0.0f;
    t.sd = 0.0f;
    if (WANT & IS_WANT_GROUND) t.gd = dpp_sub<4>(my.G, R1);
    if (WANT & IS_WANT_SKY) t.sd = dpp_sub<5>(my.K, R1);
    float mean;
    if (HAS_INVALID) {
        const float valid_dif = dpp_sub<7>(my.V, R1);
        const float sdif = dpp_sub<6>(my.S, R1);
        if (rcp_tab != nullptr) mean = mean_valid_fast(sdif, valid_dif, rcp_tab);
        else mean = (valid_dif == 0) ? 0 : sdif / valid_dif;
    } else {
        mean = fast_div(dpp_sub<6>(my.S, R1), height, r);
    }
    t.fni = (int)min(cvt_u32_sat(mean), (unsigned)(D - 1));
    t.mean = __builtin_fmaxf(mean, 0.0f);
    return t;
}

/* ====================================================================================== */
/* Wave-private LDS rings filled by LDS-DMA (is_k_unary_fast.hip, k_pw_phase1_ring)         */
/* ====================================================================================== */
#ifndef ISF_RING
#define ISF_RING 2     /* slots per wave: prefetch distance in steps (unary).  3 until the diagonal quarters (round 5): the first slots are filled while the wave walks its diagonal block in LDS; two measured equal to three (3.67 / 3.68 ms per 64 frames) and, with the unpadded records, leave the LDS of a seventh windowed workgroup per CU (is_k_unary_fast.hip) */
#endif
#define ISF_REC_F 32   /* floats of a record slot */

typedef __attribute__((address_space(3))) void* isf_lds_t;
typedef const __attribute__((address_space(1))) void* isf_glb_t;


/* s_waitcnt vmcnt(n), expcnt / lgkmcnt untouched (gfx9 encoding) */
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (7 << 4) | (0xF << 8));
    asm volatile("" ::: "memory"); /* the LDS reads of the slot stay behind the wait */
}

/* One LDS-DMA instruction: lane l's dword at gaddr goes to LDS byte address lds_base + 4 l.
 * Written as inline assembly on purpose: for `__builtin_amdgcn_global_load_lds` the compiler
 * inserts s_waitcnt vmcnt(0) in front of every LDS read that might alias the target -- here all of
 * them --, which would wait for the youngest prefetch at every step and serialise the ring.  The
 * waits are placed by hand instead (wait_vmcnt). */
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm" /* m0 is the DMA's LDS base: clobbered on purpose */
__device__ __forceinline__ void dma_dword(const float* gaddr, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dword %0, off"
                 :
                 : "v"(gaddr), "s"(lds_base)
                 : "memory", "m0");
}
#pragma clang diagnostic pop

__device__ __forceinline__ unsigned lds_addr(const float* p) {
    return __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(isf_lds_t)p);
}

__device__ __forceinline__ RowRec lds_rec(const float* p) {
    RowRec r;
    const float4* s = reinterpret_cast<const float4*>(p);
    float4* d = reinterpret_cast<float4*>(&r);
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = s[i];
    return r;
}


#endif /* IS_KERNELS_H_ */
