/* is_k_unary.hip -- the unary column DP.  See is_kernels.h. */
#include "is_kernels.h"

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


/* ====================================================================================== */
/* FAST columns: descending vB with an exact branch-and-bound (DESIGN.md "Pruning")        */
/* ====================================================================================== */
/* A wave walks its vB values DOWNWARDS.  After a full step at vB, every candidate vB' <= vB of
 * lane vT costs at least
 *     LB_o = fl(fl(sw * min(f_on, fl(f_oi - E2))) - E1o)          (object)
 *     LB_g = fl(fl(sw * f_g) - E1g),  LB_s = fl(fl(sw * f_sky) - E1s)
 * where f_* are the class-group minima of the segment (vB, vT) just evaluated: they are exact
 * integers that can only grow when the segment grows (class values >= 0 in FAST columns), the
 * terms left out are >= 0 (nic, pw / h) or >= -E (data terms, ic: PruneRec), and fp32 addition /
 * multiplication by a non-negative constant are monotone under round-to-nearest, so the bound
 * holds for the COMPUTED costs, not just the real-valued ones.  Once LB > best in every lane and
 * every type that can still receive candidates, the wave stops: nothing below can win, and nothing
 * below can tie (ties go to the smaller vB, hence the strict >). */
template <bool HAS_INVALID, bool SKY, bool DIAG, bool FIRST, int NR, bool NOGROUND>
__device__ __forceinline__ SegTerms unary_step_desc(const DevParams& P, const RowRec& my,
                                                    const RowRec& rb, const LutRow<NR>& lrow,
                                                    const float* my_tile, const float* s_rcp, int vT,
                                                    int vTc, int vhor, int vB, float hf_full,
                                                    bool row_ok, UnaryBest& b) {
    const int h = vTc + 1 - vB;
    const bool live = DIAG ? ((h > 0) && row_ok) : row_ok;
    const int hc = DIAG ? max(h, 1) : h;
    const float r = s_rcp[hc];
    const float hf = (DIAG || FIRST) ? (float)hc : hf_full;
    const SegTerms t = eval_segment<true, HAS_INVALID>(my, rb, hf, r, P.D, P.iw);
    const float od = my_tile[t.fni] - pick_lut<NR>(lrow, t.fni);
    const float pwih = P.pw * r;
    const float cost_o = P.dw * od + pwih + P.sw * t.seg_o;
    constexpr bool ALL_LANES = IS_CMPX_UPDATE && !DIAG && !FIRST;
    if (ALL_LANES) {
        take_if_le(b.o, b.vo, cost_o, vB);
    } else {
        const bool uo = live && (cost_o <= b.o);
        b.o = uo ? cost_o : b.o;
        b.vo = uo ? vB : b.vo;
    }
    if (SKY) {
        const float cost_s = P.dw * t.sd + pwih + P.sw * t.seg_s;
        if (ALL_LANES) {
            take_if_le(b.s, b.vs, cost_s, vB);
        } else {
            const bool us = live && (cost_s <= b.s);
            b.s = us ? cost_s : b.s;
            b.vs = us ? vB : b.vs;
        }
    } else if (!NOGROUND) {
        const float cost_g = P.dw * t.gd + pwih + P.sw * t.seg_g;
        if (ALL_LANES) {
            take_if_le(b.g, b.vg, cost_g, vB);
        } else {
            const bool ug = (FIRST ? (live && (vT <= vhor)) : live) && (cost_g <= b.g);
            b.g = ug ? cost_g : b.g;
            b.vg = ug ? vB : b.vg;
        }
    }
    return t;
}

struct PruneVals { /* register copy of a PruneRec + the lanes that need no bound */
    float E1o, E1g, E1s, E2;
    unsigned long long dead;  /* lanes with vT >= H: never stored                         */
    unsigned long long gdead; /* dead, or the ground data term of the lane is +inf for good */
};

template <bool SKY, bool NOGROUND>
__device__ __forceinline__ bool nothing_below_can_win(const DevParams& P, const PruneVals& pv,
                                                      const SegTerms& t, const UnaryBest& b) {
    const float lb_o = P.sw * __builtin_fminf(t.f_on, t.f_oi - pv.E2) - pv.E1o;
    unsigned long long ok = __builtin_amdgcn_ballot_w64(lb_o > b.o) | pv.dead;
    if (SKY) {
        const float lb_s = P.sw * t.f_sky - pv.E1s;
        ok &= __builtin_amdgcn_ballot_w64(lb_s > b.s) | pv.dead;
    } else if (!NOGROUND) {
        const float lb_g = P.sw * t.f_g - pv.E1g;
        ok &= __builtin_amdgcn_ballot_w64(lb_g > b.g) | pv.gdead;
    }
    return ok == ~0ull;
}

/* steps vB, vB - nw, ... >= lower; returns the next vB, or INT_MIN once the wave is done */
#define IS_WAVE_DONE (-0x40000000)
template <bool HAS_INVALID, bool SKY, bool DIAG, bool PRUNE, int NR, bool NOGROUND>
__device__ __forceinline__ int unary_range_desc(const DevParams& P, const RowRec& my,
                                                const RowRec* __restrict__ rcol,
                                                const float* __restrict__ lcol, const float* my_tile,
                                                const float* s_rcp, int vT, int vTc, int vhor, int vB,
                                                int nw, int lower, bool row_ok, int lane4,
                                                __amdgpu_buffer_rsrc_t lrsrc, LutRow<NR>& next_row,
                                                const PruneVals& pv, UnaryBest& b) {
    float hf = (float)(vTc + 1 - vB);
    const float nwf = (float)nw;
    for (; vB >= lower; vB -= nw) {
        const RowRec cur = sload_rec(rcol + vB);
        const LutRow<NR> row = next_row;
        load_lut_row<NR>(next_row, lrsrc, lcol, max(vB - nw, 0), P.D, lane4);
        const SegTerms t = unary_step_desc<HAS_INVALID, SKY, DIAG, false, NR, NOGROUND>(
            P, my, cur, row, my_tile, s_rcp, vT, vTc, vhor, vB, hf, row_ok, b);
        hf += nwf;
        if (PRUNE && IS_PRUNE && nothing_below_can_win<SKY, NOGROUND>(P, pv, t, b)) return IS_WAVE_DONE;
    }
    return vB;
}

template <bool HAS_INVALID, int NR>
__device__ __forceinline__ void unary_loop_desc(const DevParams& P, const RowRec& my,
                                                const RowRec* __restrict__ rcol,
                                                const float* __restrict__ lcol, const float* my_tile,
                                                const float* s_rcp, int vT, int vTc, int vhor, int w,
                                                int nw, int tile_lo, int vB_end, int lane4,
                                                __amdgpu_buffer_rsrc_t lrsrc, const PruneRec* prec,
                                                UnaryBest& b) {
    const bool row_ok = vT < P.H;
    if (w > vB_end) return;
    int vB = vB_end - (vB_end - w) % nw; /* the wave's largest vB: vB == w (mod nw) */
    PruneVals pv;
    {
        cprune_t q = (cprune_t)prec;
        pv.E1o = q->E1o; pv.E1g = q->E1g; pv.E1s = q->E1s; pv.E2 = q->E2;
        pv.dead = ~__builtin_amdgcn_ballot_w64(row_ok);
        pv.gdead = pv.dead | __builtin_amdgcn_ballot_w64(my.G == IS_INF);
    }
    LutRow<NR> next_row;
    load_lut_row<NR>(next_row, lrsrc, lcol, vB, P.D, lane4);
    const bool nog = IS_SKIP_GROUND_ABOVE_HORIZON && tile_lo >= vhor;
#define IS_RANGE(SKY, DIAG, PRUNE, NOG, lower)                                                     \
    vB = unary_range_desc<HAS_INVALID, SKY, DIAG, PRUNE, NR, NOG>(P, my, rcol, lcol, my_tile, s_rcp,   \
                                                                  vT, vTc, vhor, vB, nw, lower, row_ok, \
                                                                  lane4, lrsrc, next_row, pv, b);  \
    if (vB == IS_WAVE_DONE) return
    /* (the horizon may lie outside the image: vhor < 0 or vhor >= H; vB = 0 is the FIRST step) */
    IS_RANGE(true, true, false, false, max(max(vhor, tile_lo) + 1, 1)); /* sky, diagonal block */
    IS_RANGE(true, false, true, false, max(vhor + 1, 1));               /* sky, whole wave live */
    IS_RANGE(false, true, false, false, max(tile_lo + 1, 1));   /* ground, diagonal block    */
    if (nog) {
        IS_RANGE(false, false, true, true, 1);                  /* ground candidates are +inf */
    } else {
        IS_RANGE(false, false, true, false, 1);
    }
#undef IS_RANGE
    if (vB == 0) { /* first segment (:481-594): ground + object */
        const RowRec cur = sload_rec(rcol);
        const LutRow<NR> row = next_row;
        if (tile_lo == 0)
            unary_step_desc<HAS_INVALID, false, true, true, NR, false>(P, my, cur, row, my_tile, s_rcp, vT,
                                                                       vTc, vhor, 0, 0.0f, row_ok, b);
        else
            unary_step_desc<HAS_INVALID, false, false, true, NR, false>(P, my, cur, row, my_tile, s_rcp, vT,
                                                                        vTc, vhor, 0, 0.0f, row_ok, b);
    }
}

#ifdef IS_ABL_PHASES
/* debug build only: cycles (s_memtime) spent by wave 0 of every workgroup pass in staging / loop /
 * wait-for-other-waves / merge+store, summed over the launch */
__device__ unsigned long long g_phase[8];
#define IS_PHASE_MARK(k)                                                                  \
    do {                                                                                  \
        const unsigned long long now__ = __builtin_readcyclecounter();                    \
        if (threadIdx.x == 0) atomicAdd(&g_phase[k], now__ - t_phase);                    \
        t_phase = now__;                                                                  \
    } while (0)
extern "C" void isk_debug_phases_old(unsigned long long* out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(g_phase));
    if (reset) {
        unsigned long long z[8] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z));
    }
}
#else
#define IS_PHASE_MARK(k)
#endif

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
                                                  const PruneRec* __restrict__ prune,
                                                  float* __restrict__ cost_table,
                                                  int32_t* __restrict__ index_table,
                                                  const int* __restrict__ n_generic,
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
    /* The launch of the generic columns is a small grid that walks all (column, tile pair) items
     * and leaves at once when the batch has no generic column (k_prepare_columns counts them):
     * a full grid of workgroups that each only read their column's flag cost 0.13 ms per 64
     * frames in dispatch alone. */
    if (!FASTCOLS && __builtin_amdgcn_readfirstlane(*n_generic) == 0) return;
    const int n_items = ((ncols + nxcd - 1) / nxcd) * nxcd * wg_per_col;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    __syncthreads(); /* the previous item's merge area aliases this item's LUT tile */
    /* (integer division runs on the VALU: pin the uniform results back into SGPRs) */
    const int xcd = item % nxcd, q = item / nxcd;
    const int wg_in_col = __builtin_amdgcn_readfirstlane(q % wg_per_col);
    const int colg = __builtin_amdgcn_readfirstlane((q / wg_per_col) * nxcd + xcd);
    if (colg >= ncols) continue;
    if ((__builtin_amdgcn_readfirstlane(col_flags[colg]) == 0) != FASTCOLS) continue;
    const int img = __builtin_amdgcn_readfirstlane(colg / P.C);
    const int vhor = __builtin_amdgcn_readfirstlane(vhor_arr[img]);

    const int tid = threadIdx.x, lane = tid & 63;
    const int nw = blockDim.x >> 6;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const RowRec* rcol = recs + (size_t)colg * (H + 1);
    const float* lcol = lutT + (size_t)colg * (H + 1) * D;
    stage_rcp(s_rcp, rcp, H, tid, (int)blockDim.x);
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
#ifdef IS_ABL_PHASES
    unsigned long long t_phase = __builtin_readcyclecounter();
#endif

    stage_lut_tile<false>(s_tile, lcol, tile_lo, H, D, tid, (int)blockDim.x);

    const int vT = tile_lo + lane;
    const int vTc = min(vT, H - 1);
    const RowRec my = load_rec(rcol + vTc + 1);
    __syncthreads();
    IS_PHASE_MARK(0);

    UnaryBest b;
    b.g = b.o = b.s = IS_INF;
    b.vg = b.vs = -1;
    b.vo = 0; /* index_table[vT*3+OBJECT] = OBJECT at vB = 0, :592 */
    const float* my_tile = s_tile + lane * DP;
    const int vB_end = min(tile_lo + IS_TILE - 1, H - 1);
#ifndef IS_ABL_NOLOOP
    if (FASTCOLS)
        unary_loop_desc<HAS_INVALID, NR>(P, my, rcol, lcol, my_tile, s_rcp, vT, vTc, vhor, w, nw, tile_lo,
                                         vB_end, lane * 4, lrsrc, prune + colg, b);
    else
        unary_loop<false, HAS_INVALID, NR>(P, my, rcol, lcol, my_tile, s_rcp, vT, vTc, vhor, w, nw, tile_lo,
                                           vB_end, lane * 4, lrsrc, b);
#endif
    IS_PHASE_MARK(1);

    /* merge the waves' partial minima: min cost, ties -> smallest vB (= first strict minimum
     * of the reference's ascending-vB loop) */
    __syncthreads();
    IS_PHASE_MARK(2);
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
        /* a row without a finite candidate keeps the initial index (the descending walk records
         * +inf candidates, the reference's strict < never does; :592 for the object type) */
        if (!(c < IS_INF)) vb = (type == IS_OBJECT) ? 0 : -1;
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
    IS_PHASE_MARK(3);
    } /* pass */
    } /* pair */
    } /* item */
}

extern "C" hipError_t isk_launch_dp_unary_fast(const DevParams*, int, const RowRec*, const float*,
                                               const float*, const int*, const int*, const PruneRec*,
                                               float*, int32_t*, unsigned long long*, const float*,
                                               const float*, hipStream_t);

extern "C" {

size_t isk_unary_lds_bytes(const DevParams* P) {
    const size_t rcp = sizeof(float) * (((size_t)P->H + 1 + 3) & ~(size_t)3);
    const size_t tile = sizeof(float) * (size_t)IS_TILE * (P->D + 1);
    const size_t merge = (size_t)IS_UNARY_WAVES * 3 * 64 * 8; /* aliases the tile after the loop */
    return rcp + (tile > merge ? tile : merge) + 16;
}

hipError_t isk_launch_dp_unary(const DevParams* P, int ncols, int nwaves, const RowRec* recs,
                               const float* lutT, const float* rcp, const int* vhor,
                               const int* col_flags, const PruneRec* prune, float* cost_table,
                               int32_t* index_table, const int* n_generic,
                               unsigned long long* counters, const float* joined, const float* cost_T,
                               hipStream_t stream) {
    /* FAST columns: the chunk-staged kernel of is_k_unary_fast.hip whenever the shape allows it;
     * then only the generic columns are left for this file's kernel */
    /* (measured on MI355X, batch 64: 8.7 ms against 9.3 ms of the tile-pair kernel below, and no
     * scratch; IS_NO_RING_KERNEL=1 selects the old kernel for comparisons) */
    const bool fast_kernel = isk_unary_fast_chunk_rows(P) > 0 && P->knob_ring_kernel != 0;
    if (fast_kernel) {
        const hipError_t e = isk_launch_dp_unary_fast(P, ncols, recs, lutT, rcp, vhor, col_flags, prune,
                                                      cost_table, index_table, counters, joined, cost_T,
                                                      stream);
        if (e != hipSuccess) return e;
    }
    const int groups = (ncols + 7) / 8;
    /* one tile pair (big, small) per workgroup: equal-length workgroups pack best; measured on
     * MI355X at batch 32: 1 pair 9.98 ms, 2 pairs 10.10 ms, 4 pairs 10.55 ms, single tiles 11.1 ms */
    const int npairs = (P->ntiles + 1) / 2;
    const int pairs_per_wg = 1;
    const int wg_per_col = (npairs + pairs_per_wg - 1) / pairs_per_wg;
    const dim3 grid(groups * 8 * wg_per_col);
    const dim3 grid_generic(grid.x < 16384u ? grid.x : 16384u); /* walks the items, see the kernel */
    const size_t lds = isk_unary_lds_bytes(P);
    /* D <= 128: the vB-side lutT row travels in two registers per lane (LutRow<2>); wider
     * tables gather per lane */
#define IS_LAUNCH_UNARY(INV, NR)                                                                   \
    do {                                                                                           \
        if (!fast_kernel)                                                                          \
            hipLaunchKernelGGL((k_dp_unary<INV, NR, true>), grid, dim3(nwaves * 64), lds, stream,  \
                               *P, ncols, recs, lutT, rcp, vhor, col_flags, prune, cost_table,     \
                               index_table, n_generic, pairs_per_wg);                              \
        hipLaunchKernelGGL((k_dp_unary<INV, NR, false>), grid_generic, dim3(nwaves * 64), lds,     \
                           stream, *P, ncols, recs, lutT, rcp, vhor, col_flags, prune, cost_table, \
                           index_table, n_generic, pairs_per_wg);                                  \
    } while (0)
    if (P->D <= 128) {
        if (P->invalid >= 0) IS_LAUNCH_UNARY(true, 2); else IS_LAUNCH_UNARY(false, 2);
    } else {
        if (P->invalid >= 0) IS_LAUNCH_UNARY(true, 0); else IS_LAUNCH_UNARY(false, 0);
    }
#undef IS_LAUNCH_UNARY
    return hipGetLastError();
}

int isk_debug_occupancy(const DevParams* P, int nwaves) {
    int nb = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_dp_unary<false, 2, true>,
                                                 nwaves * 64, isk_unary_lds_bytes(P));
    return nb;
}

hipError_t isk_set_lds_unary(const DevParams* P) {
    hipError_t e;
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
    return isk_set_lds_unary_fast(P);
}

} /* extern "C" */
