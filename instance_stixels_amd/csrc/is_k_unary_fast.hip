/* is_k_unary_fast.hip -- unary column DP of FAST columns, built for the pruned regime.
 *
 * With the exact branch-and-bound on vB (DESIGN.md "Pruning") a (column, 64-row tile) work item
 * shrinks to its 64 diagonal steps plus a few dozen steps below the tile, so the kernel is no
 * longer bound by VALU issue but by the latency of what every step fetches.  k_dp_unary (the
 * kernel of the generic columns, is_k_unary.hip) takes the vB-side record through scalar loads
 * and the vB-side lutT row through a buffer load per wave and step: ~1.4 us of exposed latency
 * per step once the other waves no longer cover it.  Here NOTHING inside the step loop touches
 * global memory:
 *
 *   - the workgroup (8 waves, one lane per vT of the tile) walks vB downwards in CHUNKS of
 *     `chunk_rows` rows; the chunk's records (128 B each) and lutT rows are staged in LDS by all
 *     512 threads with 16-byte loads, double buffered: chunk k+1 is in flight (registers) while
 *     chunk k is evaluated, one barrier per chunk;
 *   - inside a chunk wave w takes vB = c_hi - w, c_hi - w - 8, ...: the record comes out of LDS
 *     with eight broadcast ds_read_b128, the two LUT values with per-lane ds_read_b32;
 *   - a wave whose bound says that nothing below can win (nothing_below_can_win) stops
 *     evaluating; the workgroup leaves the chunk loop when all eight have stopped.
 *
 * LDS per workgroup at 1024 x 128: 4 KB (1/h) + 33 KB (vT tile) + 2 x 20.5 KB (chunks) = 78 KB,
 * two workgroups per CU, 128 VGPRs, no scratch.
 */
#include "is_kernels.h"

#define ISF_WAVES 8
#define ISF_THREADS (ISF_WAVES * 64)
#define ISF_MAXQ 4 /* float4 per thread of one chunk's lutT rows: chunk_rows * D <= 8192 */
/* (the kernels take the actual count NQ <= ISF_MAXQ as a template parameter: registers) */

struct UnaryBestF {
    float g, o, s;
    int vg, vo, vs;
};

__device__ __forceinline__ RowRec lds_rec(const float* p) {
    RowRec r;
    const float4* s = reinterpret_cast<const float4*>(p);
    float4* d = reinterpret_cast<float4*>(&r);
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = s[i];
    return r;
}

/* One (vB, vT) evaluation; semantics of unary_step (is_k_unary.hip) with the `<=` update of a
 * descending walk.  lrow: the lutT row of vB in LDS. */
template <bool HAS_INVALID, bool SKY, bool DIAG, bool FIRST, bool NOGROUND>
__device__ __forceinline__ SegTerms fast_step(const DevParams& P, const RowRec& my, const RowRec& rb,
                                              const float* lrow, const float* my_tile,
                                              const float* s_rcp, int vT, int vTc, int vhor, int vB,
                                              bool row_ok, UnaryBestF& b) {
    const int h = vTc + 1 - vB;
    const bool live = DIAG ? ((h > 0) && row_ok) : row_ok;
    const int hc = DIAG ? max(h, 1) : h;
    const float r = s_rcp[hc]; /* RN(1/h) = (float)(1./h) = inverse_height, :485, :608 */
    const SegTerms t = eval_segment<true, HAS_INVALID>(my, rb, (float)hc, r, P.D, P.iw);
    const float od = my_tile[t.fni] - lrow[t.fni];
    const float pwih = P.pw * r;
    /* cost = dw*data + pw*(1/h) + sw*seg, left to right (:716-719, 762-765, 820-823) */
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

struct PruneValsF {
    float E1o, E1g, E1s, E2;
    unsigned long long dead;  /* lanes with vT >= H: never stored                            */
    unsigned long long gdead; /* dead, or the ground data term of the lane is +inf for good   */
};

/* see the proof sketch at unary_step_desc (is_k_unary.hip) and DESIGN.md "Pruning" */
template <bool SKY, bool NOGROUND>
__device__ __forceinline__ bool fast_nothing_below(const DevParams& P, const PruneValsF& pv,
                                                   const SegTerms& t, const UnaryBestF& b) {
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

/* registers of one chunk in flight */
template <int NQ>
struct ChunkRegs {
    float4 rows[NQ];
    float4 rec;
};

template <int NQ>
__device__ __forceinline__ void chunk_fetch(ChunkRegs<NQ>& cr, const float* __restrict__ lcol,
                                            const RowRec* __restrict__ rcol, int c_lo, int n_rows,
                                            int D, int tid) {
    const int n4 = (n_rows * D) >> 2; /* D % 4 == 0 */
    const float4* src = reinterpret_cast<const float4*>(lcol + (size_t)c_lo * D);
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const int i = tid + q * ISF_THREADS;
        if (i < n4) cr.rows[q] = src[i];
    }
    if (tid < n_rows * 8) cr.rec = reinterpret_cast<const float4*>(rcol + c_lo)[tid];
}

template <int NQ>
__device__ __forceinline__ void chunk_store(const ChunkRegs<NQ>& cr, float* b_rows, float* b_recs,
                                            int n_rows, int D, int tid) {
    const int DP = D + 1;
    const int n4 = (n_rows * D) >> 2;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const int i = tid + q * ISF_THREADS;
        if (i < n4) {
            const int e = i << 2;
            const int r = e / D, f = e - r * D;
            float* d = b_rows + r * DP + f;
            d[0] = cr.rows[q].x; d[1] = cr.rows[q].y; d[2] = cr.rows[q].z; d[3] = cr.rows[q].w;
        }
    }
    if (tid < n_rows * 8) reinterpret_cast<float4*>(b_recs)[tid] = cr.rec;
}

#ifdef IS_ABL_PHASES
__device__ unsigned long long g_fphase[8];
#define ISF_MARK(k)                                                                       \
    do {                                                                                  \
        const unsigned long long now__ = __builtin_readcyclecounter();                    \
        if (threadIdx.x == 0) atomicAdd(&g_fphase[k], now__ - t_phase);                   \
        t_phase = now__;                                                                  \
    } while (0)
#define ISF_COUNT(k) do { if (threadIdx.x == 0) atomicAdd(&g_fphase[k], 1ull); } while (0)
extern "C" void isk_debug_phases(unsigned long long* out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fphase), sizeof(g_fphase));
    if (reset) {
        unsigned long long z[8] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fphase), z, sizeof(z));
    }
}
#else
#define ISF_MARK(k)
#define ISF_COUNT(k)
#endif

template <bool HAS_INVALID, int NQ>
__global__ __launch_bounds__(ISF_THREADS, 4) void k_dp_unary_fast(
    const DevParams P, int ncols, const RowRec* __restrict__ recs, const float* __restrict__ lutT,
    const float* __restrict__ rcp, const int* __restrict__ vhor_arr,
    const int* __restrict__ col_flags, const PruneRec* __restrict__ prune,
    float* __restrict__ cost_table, int32_t* __restrict__ index_table, int chunk_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = P.H, D = P.D;
    const int DP = D + 1;
    const int CR = chunk_rows;
    float* s_rcp = (float*)smem;                       /* [H+1 -> x4]                        */
    float* s_tile = s_rcp + ((H + 1 + 3) & ~3);        /* [64][D+1] lutT rows tile_lo+1 ..    */
    float* s_buf = s_tile + ((IS_TILE * DP + 3) & ~3); /* 2 x { rows [CR][D+1], recs [CR][32] } */
    const int buf_floats = ((CR * DP + 3) & ~3) + CR * 32;

    /* XCD-aware order: blocks b, b+8, ... share an XCD/L2; the tiles of a column stay on one XCD
     * (they fetch the same lutT rows) and the tallest tiles start first */
    const int nxcd = 8;
    const int xcd = blockIdx.x % nxcd, q = blockIdx.x / nxcd;
    const int tile = __builtin_amdgcn_readfirstlane(P.ntiles - 1 - q % P.ntiles);
    const int colg = __builtin_amdgcn_readfirstlane((q / P.ntiles) * nxcd + xcd);
    if (colg >= ncols) return;
    if (__builtin_amdgcn_readfirstlane(col_flags[colg]) != 0) return; /* generic column: k_dp_unary */
    const int vhor = __builtin_amdgcn_readfirstlane(vhor_arr[colg / P.C]);

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const RowRec* rcol = recs + (size_t)colg * (H + 1);
    const float* lcol = lutT + (size_t)colg * (H + 1) * D;
    const int tile_lo = tile * IS_TILE;
    const int vT = tile_lo + lane;
    const int vTc = min(vT, H - 1);
    const bool row_ok = vT < H;
    const int vB_end = min(tile_lo + IS_TILE - 1, H - 1);

#ifdef IS_ABL_PHASES
    unsigned long long t_phase = __builtin_readcyclecounter();
#endif
    /* ---- prologue: 1/h table, the tile's lutT rows, this lane's record, the first chunk */
    ChunkRegs<NQ> cr;
    int c_hi = vB_end, c_lo = max(c_hi - CR + 1, 0);
    chunk_fetch(cr, lcol, rcol, c_lo, c_hi - c_lo + 1, D, tid);
    const RowRec my = load_rec(rcol + vTc + 1);
    for (int i = tid; i <= H; i += ISF_THREADS) s_rcp[i] = rcp[i];
    stage_lut_tile<true>(s_tile, lcol, tile_lo, H, D, tid, ISF_THREADS);
    chunk_store(cr, s_buf, s_buf + ((CR * DP + 3) & ~3), c_hi - c_lo + 1, D, tid);

    PruneValsF pv;
    {
        cprune_t pq = (cprune_t)(prune + colg);
        pv.E1o = pq->E1o; pv.E1g = pq->E1g; pv.E1s = pq->E1s; pv.E2 = pq->E2;
        pv.dead = ~__builtin_amdgcn_ballot_w64(row_ok);
        pv.gdead = pv.dead | __builtin_amdgcn_ballot_w64(my.G == IS_INF);
    }
    UnaryBestF b;
    b.g = b.o = b.s = IS_INF;
    b.vg = b.vs = -1;
    b.vo = 0; /* index_table[vT*3+OBJECT] = OBJECT at vB = 0, :592 */
    const float* my_tile = s_tile + lane * DP;
    const bool nog = IS_SKIP_GROUND_ABOVE_HORIZON && tile_lo >= vhor;
    bool done = false; /* wave-uniform: nothing below can win any more */
    __syncthreads();
    ISF_MARK(0);

    /* ---- chunk loop, vB downwards */
    for (int k = 0;; k++) {
        const int n_lo = max(c_lo - CR, 0); /* next chunk = [n_lo, c_lo - 1] */
        const bool more = c_lo > 0;
        if (more) chunk_fetch(cr, lcol, rcol, n_lo, c_lo - n_lo, D, tid);

        const float* b_rows = s_buf + (k & 1) * buf_floats;
        const float* b_recs = b_rows + ((CR * DP + 3) & ~3);
        if (!done) {
            for (int vB = c_hi - w; vB >= c_lo; vB -= ISF_WAVES) {
                const int j = vB - c_lo;
#ifdef IS_FAST_SREC
                const RowRec rb = sload_rec(rcol + vB);
#else
                const RowRec rb = lds_rec(b_recs + j * 32);
#endif
                const float* lrow = b_rows + j * DP;
                const bool diag = vB > tile_lo;
                if (vB == 0) { /* first segment (:481-594): ground + object */
                    if (diag)
                        fast_step<HAS_INVALID, false, true, true, false>(P, my, rb, lrow, my_tile, s_rcp, vT,
                                                                         vTc, vhor, 0, row_ok, b);
                    else
                        fast_step<HAS_INVALID, false, false, true, false>(P, my, rb, lrow, my_tile, s_rcp, vT,
                                                                          vTc, vhor, 0, row_ok, b);
                } else if (vB > vhor) { /* vB - 1 >= vhor: sky + object (:729) */
                    if (diag) {
                        fast_step<HAS_INVALID, true, true, false, false>(P, my, rb, lrow, my_tile, s_rcp, vT,
                                                                         vTc, vhor, vB, row_ok, b);
                    } else {
                        const SegTerms t = fast_step<HAS_INVALID, true, false, false, false>(
                            P, my, rb, lrow, my_tile, s_rcp, vT, vTc, vhor, vB, row_ok, b);
                        if (IS_PRUNE && fast_nothing_below<true, false>(P, pv, t, b)) { done = true; break; }
                    }
                } else { /* ground + object (:687) */
                    if (diag) {
                        fast_step<HAS_INVALID, false, true, false, false>(P, my, rb, lrow, my_tile, s_rcp, vT,
                                                                          vTc, vhor, vB, row_ok, b);
                    } else if (nog) {
                        const SegTerms t = fast_step<HAS_INVALID, false, false, false, true>(
                            P, my, rb, lrow, my_tile, s_rcp, vT, vTc, vhor, vB, row_ok, b);
                        if (IS_PRUNE && fast_nothing_below<false, true>(P, pv, t, b)) { done = true; break; }
                    } else {
                        const SegTerms t = fast_step<HAS_INVALID, false, false, false, false>(
                            P, my, rb, lrow, my_tile, s_rcp, vT, vTc, vhor, vB, row_ok, b);
                        if (IS_PRUNE && fast_nothing_below<false, false>(P, pv, t, b)) { done = true; break; }
                    }
                }
            }
        }
        ISF_MARK(1);
        ISF_COUNT(5);
        if (!more) break;
        /* the other buffer was last read while chunk k-1 was evaluated, i.e. before the barrier
         * that ended iteration k-1 */
        float* n_rows_p = s_buf + ((k + 1) & 1) * buf_floats;
        chunk_store(cr, n_rows_p, n_rows_p + ((CR * DP + 3) & ~3), c_lo - n_lo, D, tid);
        c_hi = c_lo - 1;
        c_lo = n_lo;
        const bool all_done = __syncthreads_and(done);
        ISF_MARK(2);
        if (all_done) break;
    }

    /* ---- merge the waves' partial minima: min cost, ties -> smallest vB */
    __syncthreads();
    float* m_cost = s_buf;                               /* [8][3][64] */
    int* m_vb = (int*)(m_cost + ISF_WAVES * 3 * 64);     /* [8][3][64] */
    m_cost[(w * 3 + 0) * 64 + lane] = b.g; m_vb[(w * 3 + 0) * 64 + lane] = b.vg;
    m_cost[(w * 3 + 1) * 64 + lane] = b.o; m_vb[(w * 3 + 1) * 64 + lane] = b.vo;
    m_cost[(w * 3 + 2) * 64 + lane] = b.s; m_vb[(w * 3 + 2) * 64 + lane] = b.vs;
    __syncthreads();
    if (tid < 3 * 64) {
        const int type = tid >> 6;
        float c = m_cost[(0 * 3 + type) * 64 + lane];
        int vb = m_vb[(0 * 3 + type) * 64 + lane];
        for (int ww = 1; ww < ISF_WAVES; ww++) {
            const float c2 = m_cost[(ww * 3 + type) * 64 + lane];
            const int vb2 = m_vb[(ww * 3 + type) * 64 + lane];
            const bool take = (c2 < c) || (c2 == c && vb2 >= 0 && (vb < 0 || vb2 < vb));
            if (take) { c = c2; vb = vb2; }
        }
        /* a row without a finite candidate keeps the initial index (the descending walk records
         * +inf candidates, the reference's strict < never does; :592 for the object type) */
        if (!(c < IS_INF)) vb = (type == IS_OBJECT) ? 0 : -1;
        s_tile[type * 64 + lane] = c; /* the tile is no longer needed */
        ((int*)s_tile)[3 * 64 + type * 64 + lane] = vb;
    }
    __syncthreads();
    if (w == 0 && row_ok) {
        const size_t o = ((size_t)colg * H + vT) * 3;
        float* cd = cost_table + o;
        int32_t* id = index_table + o;
        const int* f_vb = (const int*)s_tile + 3 * 64;
        cd[0] = s_tile[0 * 64 + lane]; cd[1] = s_tile[1 * 64 + lane]; cd[2] = s_tile[2 * 64 + lane];
        id[0] = f_vb[0 * 64 + lane]; id[1] = f_vb[1 * 64 + lane]; id[2] = f_vb[2 * 64 + lane];
    }
    ISF_MARK(3);
}

extern "C" {

size_t isk_unary_fast_lds_bytes(const DevParams* P, int chunk_rows) {
    const size_t DP = (size_t)P->D + 1;
    const size_t rcp = ((size_t)P->H + 1 + 3) & ~(size_t)3;
    const size_t tile = ((size_t)IS_TILE * DP + 3) & ~(size_t)3;
    const size_t buf = (((size_t)chunk_rows * DP + 3) & ~(size_t)3) + (size_t)chunk_rows * 32;
    size_t two = 2 * buf;
    const size_t merge = (size_t)ISF_WAVES * 3 * 64 * 2;
    if (two < merge) two = merge;
    return sizeof(float) * (rcp + tile + two) + 16;
}

/* Largest chunk (rows) the shape allows: D % 4 == 0 (16-byte row loads), a chunk's rows fit the
 * per-thread registers, and two workgroups (or at least one) fit a CU's 160 KiB of LDS.
 * 0 = the shape cannot use this kernel. */
int isk_unary_fast_chunk_rows(const DevParams* P) {
    if ((P->D & 3) != 0) return 0;
    for (int pass = 0; pass < 2; pass++) {
        const size_t limit = pass == 0 ? 80 * 1024 : 160 * 1024;
        for (int cr = 32; cr >= 8; cr >>= 1) {
            if ((size_t)cr * P->D > (size_t)ISF_MAXQ * 4 * ISF_THREADS) continue;
            if (isk_unary_fast_lds_bytes(P, cr) <= limit) return cr;
        }
    }
    return 0;
}

hipError_t isk_set_lds_unary_fast(const DevParams* P) {
    const int cr = isk_unary_fast_chunk_rows(P);
    if (cr == 0) return hipSuccess;
    const int b = (int)isk_unary_fast_lds_bytes(P, cr);
    hipError_t e = hipSuccess;
#define ISF_SET(INV, NQ)                                                                          \
    if (e == hipSuccess)                                                                          \
    e = hipFuncSetAttribute((const void*)k_dp_unary_fast<INV, NQ>,                                \
                            hipFuncAttributeMaxDynamicSharedMemorySize, b)
    ISF_SET(true, 2); ISF_SET(false, 2); ISF_SET(true, ISF_MAXQ); ISF_SET(false, ISF_MAXQ);
#undef ISF_SET
    return e;
}

/* FAST columns of the batch; the caller runs k_dp_unary<.., false> for the generic ones. */
hipError_t isk_launch_dp_unary_fast(const DevParams* P, int ncols, const RowRec* recs,
                                    const float* lutT, const float* rcp, const int* vhor,
                                    const int* col_flags, const PruneRec* prune, float* cost_table,
                                    int32_t* index_table, hipStream_t stream) {
    const int cr = isk_unary_fast_chunk_rows(P);
    const int groups = (ncols + 7) / 8;
    const dim3 grid(groups * 8 * P->ntiles);
    const size_t lds = isk_unary_fast_lds_bytes(P, cr);
    const bool small = (size_t)cr * P->D <= (size_t)2 * 4 * ISF_THREADS;
#define ISF_LAUNCH(INV, NQ)                                                                       \
    hipLaunchKernelGGL((k_dp_unary_fast<INV, NQ>), grid, dim3(ISF_THREADS), lds, stream, *P, ncols,  \
                       recs, lutT, rcp, vhor, col_flags, prune, cost_table, index_table, cr)
    if (P->invalid >= 0) {
        if (small) ISF_LAUNCH(true, 2); else ISF_LAUNCH(true, ISF_MAXQ);
    } else {
        if (small) ISF_LAUNCH(false, 2); else ISF_LAUNCH(false, ISF_MAXQ);
    }
#undef ISF_LAUNCH
    return hipGetLastError();
}

} /* extern "C" */
