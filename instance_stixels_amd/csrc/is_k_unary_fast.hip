/* is_k_unary_fast.hip -- unary column DP of FAST columns, built for the pruned regime.
 *
 * With the exact branch-and-bound on vB (DESIGN.md "Pruning") a (column, 64-row tile) work item
 * shrinks to its 64 diagonal steps plus a few dozen steps below the tile, so what a step
 * fetches must never stall it (with that achieved the kernel is bound by VALU issue again: 80-86 %
 * of the issue slots, DESIGN.md sections 6-7).  k_dp_unary (the
 * kernel of the generic columns, is_k_unary.hip) takes the vB-side record through scalar loads
 * and the vB-side lutT row through a buffer load per wave and step: ~1 us of exposed latency
 * per step once the other waves no longer cover it.  Here NOTHING inside the step loop waits for
 * global memory:
 *
 *   - a wave (one lane per vT of the tile) walks its vB values downwards and owns a RING of K LDS
 *     slots, each holding the lutT row and the 128-byte record of one vB;
 *   - the slots are filled by `global_load_lds` (LDS-DMA: no VGPRs, no waiting), K steps ahead
 *     of their use; `s_waitcnt vmcnt(NV * (K - 1))` before a step guarantees that its own slot
 *     has landed while the K - 1 younger prefetches stay in flight (loads return in order);
 *   - the 128-byte record of vB is never copied into 32 VGPRs: a lane reads two dwords of the slot
 *     (R0 = rec[l & 15], R1 = rec[16 + (l & 15)]) and every subtraction takes its vB operand as a
 *     DPP `row_newbcast:k` of R0 / R1 (eval_segment_dpp, is_kernels.h); the two LUT values are
 *     per-lane ds_read_b32; no barrier inside the loop: the waves of a workgroup only share the
 *     tile's lutT rows (vT side), the 1/h table and the final merge;
 *   - a wave whose bound says that nothing below can win leaves its loop; once the object bound
 *     holds, the rest of its walk evaluates only the ground or the sky candidates, sixteen per round trip (gs_walk).
 *
 * LDS per workgroup at 1024 x 128: 4 KB (1/h) + 33 KB (vT tile, reused by the merge) + 8 waves x 3
 * slots x (512 + 128) B (rings) = 52 KB: three workgroups = 24 waves per CU at 74 VGPRs, no scratch.
 *
 * Round 4, the WIN instantiation (fn windows, is_device.h): the tile and the ring slots hold the 32 lutT
 * columns from win_lo[column][tile] on instead of all D -- 4 KB (1/h) + 8.4 KB (tile) + 4 waves x 3 slots x
 * (128 + 128) B = 15.5 KB: SEVEN 4-wave workgroups per CU (the registers' limit), a slot fill is ONE LDS-DMA
 * instruction (lanes 0-31 the row window, 32-63 the record); a lane whose floor(mean) lies outside the window
 * reads global memory.  Every tile of calls of >= 2048 columns runs this way: DP 5.75 -> 4.57 ms per 64 frames.
 *
 * Round 5: the ground- / sky-only tail of a walk in batches of sixteen candidates per memory round trip (gs_walk); the
 * diagonal block of a windowed tile in 16-lane quarters on operands staged in LDS (diag_quarters: ten steps instead
 * of sixteen; + 8 KB for the tile's 64 records, a ring of two slots = 22.8 KB per workgroup, seven per CU): 3.67 ms per 64 frames, and
 * the windowed launch at every call size (ISF_WIN_MIN_COLS).
 */
#include "is_kernels.h"

#ifndef ISF_WAVES
#define ISF_WAVES 8
#endif
#ifndef ISF_DPP
#define ISF_DPP 1 /* the record of vB as two dwords per lane + DPP operands instead of 32 VGPRs */
#endif
#ifndef ISF_GS_STEPS
#define ISF_GS_STEPS 1 /* ground / sky-only steps once the object bound holds for the wave */
#endif
#ifndef ISF_OCC_INV
#define ISF_OCC_INV 6 /* with an invalid-disparity value (round 5: the mean through mean_valid_fast; 5 while it was an IEEE division; 7 until round 5: at 72 VGPRs these instantiations spill 2-6 registers; measured again with the 22.8 KB workgroups that LDS admits seven of: 6 | 7 waves per SIMD 10 600 | 10 220 frames/s at invalid_disparity = 0, 15 550 | 15 130 on the 784x1792 crop) */
#endif
#ifndef ISF_OCC
#define ISF_OCC 7 /* waves per SIMD the kernel is compiled for: 68 VGPRs without spills; LDS keeps three workgroups = 6 per SIMD resident (7 measured 0.5-1 % faster than 6) */
#endif
#ifndef ISF_OCC_GEN
#define ISF_OCC_GEN 5 /* the carry-only instantiation: the 32-value network of a rebuild next to the lane's 32-register record (7: 35 spilled VGPRs, the walk 3.6 x slower) */
#endif
#ifndef ISF_SREC
#define ISF_SREC 1 /* the class-prefix half of the vB record as scalar operands (eval_segment_mix) */
#endif
#define ISF_THREADS (ISF_WAVES * 64)
#ifndef ISF_WIN_WAVES
#define ISF_WIN_WAVES 4 /* waves per workgroup of the windowed tiles */
#endif
static_assert(ISF_WIN_WAVES >= 3 && ISF_WIN_WAVES <= ISF_WAVES, "the merge runs on 3 x 64 threads (one wave per type)");
#ifndef ISF_WIN_MIN_COLS
#define ISF_WIN_MIN_COLS 0 /* columns per call from which the windowed launch is used.  Round 4: 2048 (frames/s windowed | classic at batch 2: 6470 | 6050, 4: 6090 | 6350, 8: 6890 | 6680, 16: 7480 | 7010, 32: 7860 | 7200); with the diagonal blocks in quarters (round 5) the windowed launch wins at every size: batch 1: 6290 | 5220, 2: 8180 | 6140, 4: 7110 | 6380, 8: 9190 | 9070 */
#endif

struct UnaryBestF {
    float g, o, s;
    int vg, vo, vs;
};

/* One (vB, vT) evaluation; semantics of unary_step (is_k_unary.hip) with the `<=` update of a
 * descending walk.  lrow: the lutT row of vB in LDS. */
/* WIN: my_tile holds the fn window [win.lo, win.lo + IS_P1_WIN) of the lane's lutT row (is_device.h); a lane
 * outside it reads global memory -- the one access of the loop that waits for memory (the compiler's
 * vmcnt(0) for it also drains the ring's prefetches: correct, and rare below the horizon). */
struct FastWin {
    int lo;
    const float* gcol; /* lutT of the column (row vB: gcol + vB * D) */
    int* misses;       /* wave-uniform count of steps with a lane outside */
    /* GEN (carry-only lutT): what lut_entry_exact needs, and the slack of the lazy test */
    const float* dcol;   /* joined disparities of the column */
    const float* cost_T; /* [dis][fn] object data costs */
    int vT1;             /* min(vT + 1, H): the lutT row of this lane */
    float E1o;           /* PruneRec.E1o: fl(dw od) >= -E1o for every segment of the column */
};

/* ====================================================================================== */
/* GEN: the lutT rows the walk reads, REBUILT from the carry rows (DevParams::lut_carry)     */
/* ====================================================================================== */
/* k_object_lut chains 32-row blocks: block k takes the per-row costs x[l] = cost_T[(int)d[32 k + l]][fn], adds
 * its carry lutT[32 k][fn] into x[0] and runs the 32-element Kogge-Stone network c[l] += c[l - j], j = 1, 2, 4,
 * 8, 16 (StixelsKernels.cu:249-272; object_lut_body, is_k_prepare.hip); c[l] is lutT[32 k + l + 1][fn] and c[31]
 * the next carry.  In carry-only calls the prepare kernel stores just the rows 32 k -- 1/32 of its 8.6 GB per 64
 * frames -- and this kernel recomputes the rows it reads with the SAME additions on the same values in the same
 * association: bit-identical rows.
 *   - the vT-side tile (rows tile_lo + 1 .. tile_lo + 64 = the outputs of the blocks 2 t and 2 t + 1, the 32
 *     window columns): wave 0 in the prologue, block 2 t in lanes 0-31, block 2 t + 1 in lanes 32-63;
 *   - the vB side: a wave visits vB = vB_top - 4 m, i.e. the rows l = (vB - 1) mod 32 of ONE residue class mod 4
 *     of every block below its tile.  It rebuilds those 8 rows of TWO blocks at a time (lanes 0-31: block k,
 *     lanes 32-63: block k - 1) into a private LDS cache that serves its next 16 steps: of the network only the
 *     nodes those 8 outputs depend on (48 additions instead of 129);
 *   - a lane whose floor(mean) falls outside the window needs an entry nobody rebuilt: first the lazy test (the
 *     candidate cannot win whatever its data term is: skipped, exactly), else lut_entry_exact. */
#ifndef ISF_GEN
#define ISF_GEN 1
#endif
#define ISF_GEN_HALF_F (8 * IS_P1_WIN + 32) /* floats of one half's 8 cached rows; + 32: the halves' stores hit disjoint banks */
#define ISF_GEN_CACHE_F (2 * ISF_GEN_HALF_F) /* per wave; its first 64 words double as the staging of the row offsets */

/* A wave-uniform pointer that arrives in VGPRs (arguments of a called function do) back into SGPRs: loads through
 * it then take the scalar-base form again. */
template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (T*)(((unsigned long long)hi << 32) | lo);
}

/* byte offset of row `row` of the column in cost_T: (int)d clamped like object_lut_body does; rows beyond the
 * image use dis = 0 (:244-247) */
__device__ __forceinline__ int gen_row_offset(const float* __restrict__ dcol, int row, int H, int D) {
    int dis = (row >= 0 && row < H) ? (int)dcol[row] : 0;
    dis = min(max(dis, 0), D - 1);
    return dis * D * (int)sizeof(float);
}

/* The outputs c[l], l = RHO, RHO + 4, .., RHO + 28, of the network on x[0..31] (x[0] already holds the carry):
 * only the nodes they depend on.  Level j keeps the positions the next level reads. */
template <int RHO>
__device__ __forceinline__ void gen_rows_of_residue(const float (&x)[32], float (&out)[8]) {
    constexpr int P1 = RHO & 1;
    float c1[16]; /* positions P1 + 2 i */
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int p = P1 + 2 * i;
        c1[i] = (p >= 1) ? x[p] + x[p - 1] : x[p];
    }
    float c2[8]; /* positions RHO + 4 i: c1[p] + c1[p - 2] */
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int p = RHO + 4 * i;
        c2[i] = (p >= 2) ? c1[(p - P1) / 2] + c1[(p - 2 - P1) / 2] : c1[(p - P1) / 2];
    }
    float c3[8]; /* c2[p] + c2[p - 4] */
#pragma unroll
    for (int i = 0; i < 8; i++) c3[i] = (RHO + 4 * i >= 4) ? c2[i] + c2[i - 1] : c2[i];
    float c4[8]; /* c3[p] + c3[p - 8] */
#pragma unroll
    for (int i = 0; i < 8; i++) c4[i] = (RHO + 4 * i >= 8) ? c3[i] + c3[i - 2] : c3[i];
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = (RHO + 4 * i >= 16) ? c4[i] + c4[i - 4] : c4[i];
}

/* The rows of residue RHO of the blocks k (lanes 0-31) and k - 1 (lanes 32-63), window columns win_lo + (lane &
 * 31), into the wave's cache: row j of a half at cache[half * ISF_GEN_HALF_F + j * IS_P1_WIN + column].  The
 * block's 32 row offsets go through the first 64 words of the cache (every row of it is dead when a rebuild
 * starts: the walk only descends), so that a lane reads the offsets of ITS half's block with plain LDS loads. */
/* (inlined: as CALLED functions the rebuilds made the compiler keep 70 values of the walk in scratch for good -- what
 * lives across a call must sit in the 32 callee-saved VGPRs of a 72-register budget)*/
template <int RHO>
__device__ __forceinline__ void gen_rebuild_rows(float* cache, const float* lcol_, const float* dcol_, const float* cost_T_,
                                              int k, int win_lo, int H, int D) {
    const float* __restrict__ lcol = uniform_ptr(lcol_);
    const float* __restrict__ dcol = uniform_ptr(dcol_);
    const float* __restrict__ cost_T = uniform_ptr(cost_T_);
    k = __builtin_amdgcn_readfirstlane(k); win_lo = __builtin_amdgcn_readfirstlane(win_lo);
    H = __builtin_amdgcn_readfirstlane(H); D = __builtin_amdgcn_readfirstlane(D);
    const int lane = (int)(threadIdx.x & 63);
    const int half = lane >> 5, c = lane & 31;
    const int kh = max(k - half, 0); /* (block -1 does not exist: recomputes block 0, never read) */
    int* s_off = (int*)cache;
    s_off[lane] = gen_row_offset(dcol, 32 * kh + c, H, D);
    const int fn = min(IS_WIN_COL(win_lo, c), D - 1);
    const unsigned fn4 = (unsigned)fn * 4u; /* (uniform base + 32-bit lane offset: one integer addition per load) */
    const float carry = lcol[(size_t)(32 * kh) * D + fn];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* (one wave: its own LDS stores are visible to its loads) */
    float x[32];
#pragma unroll
    for (int p = 0; p < 32; p++)
        x[p] = *(const float*)((const char*)cost_T + (size_t)((unsigned)s_off[half * 32 + p] + fn4));
    x[0] += carry; /* :249-251 */
    float out[8];
    gen_rows_of_residue<RHO>(x, out);
    asm volatile("" ::: "memory");
    float* dst = cache + half * ISF_GEN_HALF_F + c;
#pragma unroll
    for (int j = 0; j < 8; j++) dst[j * IS_P1_WIN] = out[j];
}

/* The vT-side tile of a carry-only call: lutT rows tile_lo + 1 .. tile_lo + 64, window columns, by ONE wave (block
 * 2 t in lanes 0-31, block 2 t + 1 in lanes 32-63); s_off: 64 words of scratch.  Rows beyond the image hold
 * values no live lane reads. */
__device__ __forceinline__ void gen_window_tile(float* s_tile, int* s_off, const float* lcol_, const float* dcol_,
                                             const float* cost_T_, int tile_lo, int win_lo, int H, int D) {
    const float* __restrict__ lcol = uniform_ptr(lcol_);
    const float* __restrict__ dcol = uniform_ptr(dcol_);
    const float* __restrict__ cost_T = uniform_ptr(cost_T_);
    tile_lo = __builtin_amdgcn_readfirstlane(tile_lo); win_lo = __builtin_amdgcn_readfirstlane(win_lo);
    H = __builtin_amdgcn_readfirstlane(H); D = __builtin_amdgcn_readfirstlane(D);
    const int lane = (int)(threadIdx.x & 63);
    constexpr int WPs = IS_P1_WIN + 1;
    const int half = lane >> 5, c = lane & 31;
    const int i0 = tile_lo + 32 * half; /* the block's first row (a multiple of 32) */
    s_off[lane] = gen_row_offset(dcol, i0 + c, H, D);
    const int fn = min(IS_WIN_COL(win_lo, c), D - 1);
    const unsigned fn4 = (unsigned)fn * 4u;
    const float carry = lcol[(size_t)min(i0, (H / 32) * 32) * D + fn]; /* (a block that starts beyond the image is never read) */
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float x[32];
#pragma unroll
    for (int p = 0; p < 32; p++)
        x[p] = *(const float*)((const char*)cost_T + (size_t)((unsigned)s_off[half * 32 + p] + fn4));
    x[0] += carry;
#pragma unroll
    for (int j = 1; j < 32; j <<= 1) {
#pragma unroll
        for (int l = 31; l >= j; l--) x[l] += x[l - j];
    }
    float* dst = s_tile + (32 * half) * WPs + c;
#pragma unroll
    for (int l = 0; l < 32; l++) dst[l * WPs] = x[l];
}

/* lutT[v][fn] of the column for ANY (v, fn), per lane, from the carry rows: the slow, exact path of the lanes
 * outside every window.  v = 32 k + l + 1: position l of block k; with y[q] = x[l - q] the network's value at
 * position l is a fixed binary tree over y[0..31] in which a node whose right half would start below position 0
 * keeps its left half (c[p] += c[p - j] only for p >= j): node i of level j covers y[i 2^j .. (i + 1) 2^j - 1] and
 * joins its halves when the left half's top position l - i 2^j is at least 2^(j - 1).  Evaluated depth first, four
 * leaves at a time: a handful of registers, but up to eight dependent pairs of loads (measured: with all 64 loads
 * in flight the function needs 40 registers more, and what lives across a CALL must sit in callee-saved
 * registers -- 36 spilled VGPRs of the walk, 11.9 instead of 8.1 ms; inlined into the seven step variants: 224). */
__device__ __noinline__ float lut_entry_exact(const float* __restrict__ lcol, const float* __restrict__ dcol,
                                              const float* __restrict__ cost_T, int v, int fn, int H, int D) {
    if ((v & 31) == 0) return lcol[(size_t)v * D + fn]; /* a carry row: materialised */
    const int k = (v - 1) >> 5, l = (v - 1) & 31;
    const float carry = lcol[(size_t)(32 * k) * D + fn];
    const char* tbase = (const char*)(cost_T + fn);
    auto y = [&](int q) -> float { /* x[l - q]; positions below 0 are read (clamped) but never used */
        const int p = l - q;
        const float xv = *(const float*)(tbase + gen_row_offset(dcol, 32 * k + max(p, 0), H, D));
        return (p == 0) ? xv + carry : xv; /* :249-251, the carry enters at position 0 */
    };
    float t5 = 0.0f;
#pragma unroll 1
    for (int i4 = 0; i4 < 2; i4++) {
        float t4 = 0.0f;
        if (i4 == 1 && !(l >= 16)) break;
#pragma unroll 1
        for (int i3 = 0; i3 < 2; i3++) {
            const int b3 = 16 * i4 + 8 * i3;
            if (i3 == 1 && !(l - 16 * i4 >= 8)) break;
            float t3 = 0.0f;
#pragma unroll
            for (int i2 = 0; i2 < 2; i2++) {
                const int b2 = b3 + 4 * i2;
                if (i2 == 1 && !(l - b3 >= 4)) break;
                const float y0 = y(b2), y1 = y(b2 + 1), y2 = y(b2 + 2), y3 = y(b2 + 3);
                const float t1a = (l - b2 >= 1) ? y0 + y1 : y0;
                const float t1b = (l - b2 - 2 >= 1) ? y2 + y3 : y2;
                const float t2 = (l - b2 >= 2) ? t1a + t1b : t1a;
                t3 = (i2 == 0) ? t2 : t3 + t2;
            }
            t4 = (i3 == 0) ? t3 : t4 + t3;
        }
        t5 = (i4 == 0) ? t4 : t5 + t4;
    }
    return t5;
}
template <bool HAS_INVALID, bool SKY, bool DIAG, bool FIRST, bool NOGROUND, bool WIN = false, bool GEN = false>
__device__ __forceinline__ SegTerms fast_step(const DevParams& P, const RowRec& my, const float* srec,
                                              const float* lrow, const float* my_tile,
                                              const float* s_rcp, int vT, int vTc, int vhor, int vB,
                                              bool row_ok, UnaryBestF& b, const isk_f16v& S,
                                              const FastWin win = FastWin()) {
    const int h = vTc + 1 - vB;
    const bool live = DIAG ? ((h > 0) && row_ok) : row_ok;
    const int hc = DIAG ? max(h, 1) : h;
    const float r = s_rcp[hc]; /* RN(1/h) = (float)(1./h) = inverse_height, :485, :608 */
#if ISF_DPP
    const int l15 = threadIdx.x & 15;
    /* FIRST (vB = 0): ground + object; otherwise the sky OR the ground candidate, or neither */
    constexpr int WANT = SKY ? IS_WANT_SKY : (NOGROUND ? 0 : IS_WANT_GROUND);
#if ISF_SREC
    const SegTerms t = eval_segment_mix<HAS_INVALID, WANT>(my, S, srec[16 + l15], (float)hc, r, P.D, P.iw, s_rcp);
#else
    const SegTerms t = eval_segment_dpp<HAS_INVALID, WANT>(my, srec[l15], srec[16 + l15], (float)hc, r, P.D,
                                                           P.iw, s_rcp);
#endif
#else
    const RowRec rb = lds_rec(srec);
    const SegTerms t = eval_segment<true, HAS_INVALID>(my, rb, (float)hc, r, P.D, P.iw, s_rcp);
#endif
    float vtv, vbv;
    if (WIN) {
        const int fo = IS_WIN_FIND(win.lo, t.fni);
        const bool inw = fo >= 0;
        /* both ends from the windows: the lane's row in the tile, the row of vB in the ring slot (the ring
         * fetches the SAME window of every vB row: 128 instead of 512 bytes per step) */
        vtv = my_tile[inw ? fo : 0];
        vbv = lrow[inw ? fo : 0];
        bool need = live && !inw; /* (the dead lanes of a diagonal step hold no segment) */
        if (GEN) {
            /* the lazy test: whatever its data term is, fl(dw od) >= -E1o (lemma L3), so by monotone rounding
             * cost_o >= fl(fl(-E1o + pw / h) + fl(sw seg_o)); a candidate whose bound already exceeds the lane's best
             * object cost cannot pass the `<=` update: it needs no table entry (its od is set to +inf below, the
             * update then fails the same way).  E1o = +inf (pruning off for the column): never skipped. */
            const float lazy = (P.pw * r - win.E1o) + P.sw * t.seg_o;
            need = need && !(lazy > b.o);
        }
        if (__builtin_amdgcn_ballot_w64(need) != 0ull) {
            if (need) {
                if (GEN) {
#ifdef ISF_GEN_ABL_NOSLOW /* timing-only ablation (wrong results): what do the exact entries cost */
                    vtv = 1.0f; vbv = 0.0f;
#else
                    vtv = lut_entry_exact(win.gcol, win.dcol, win.cost_T, win.vT1, t.fni, P.H, P.D);
                    vbv = lut_entry_exact(win.gcol, win.dcol, win.cost_T, vB, t.fni, P.H, P.D);
#endif
                } else {
                    vtv = (win.gcol + (size_t)win.vT1 * P.D)[(unsigned)t.fni]; /* (the address on the spot: a per-lane pointer held through the walk cost two VGPRs, a spill reloaded in every step) */
                    vbv = (win.gcol + (size_t)vB * P.D)[(unsigned)t.fni];
                }
            }
            (*win.misses)++;
        }
        if (GEN && live && !inw && !need) vtv = IS_INF, vbv = 0.0f; /* skipped: cost_o = +inf, never taken (b.o is finite: lazy > b.o) */
    } else {
        vtv = my_tile[t.fni];
        vbv = lrow[t.fni];
    }
    const float od = vtv - vbv;
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
    float E1o, E1g, E1s, E2; /* E2: three times PruneRec.E2, see seg_o_lower_bound */
    unsigned long long dead;  /* lanes with vT >= H: never stored                            */
    unsigned long long gdead; /* dead, or the ground data term of the lane is +inf for good   */
};

/* see the proof sketch at unary_step_desc (is_k_unary.hip) and DESIGN.md "Pruning".  Bit 0: the
 * object bound holds in every lane, bit 1: the ground / sky bound does. */
template <bool SKY, bool NOGROUND>
__device__ __forceinline__ int fast_bounds(const DevParams& P, const PruneValsF& pv, const SegTerms& t,
                                           const UnaryBestF& b) {
    const float lb_o = P.sw * seg_o_lower_bound(t, pv.E2) - pv.E1o; /* (pv.E2 holds 3 * PruneRec.E2) */
    const bool ok_o = (__builtin_amdgcn_ballot_w64(lb_o > b.o) | pv.dead) == ~0ull;
    bool ok_x = true;
    if (SKY) {
        const float lb_s = P.sw * t.seg_s - pv.E1s;
        ok_x = (__builtin_amdgcn_ballot_w64(lb_s > b.s) | pv.dead) == ~0ull;
    } else if (!NOGROUND) {
        const float lb_g = P.sw * t.seg_g - pv.E1g;
        ok_x = (__builtin_amdgcn_ballot_w64(lb_g > b.g) | pv.gdead) == ~0ull;
    }
    return (ok_o ? 1 : 0) | (ok_x ? 2 : 0);
}

/* The ground- / sky-only candidates of a wave whose object type is closed, SIXTEEN vB per memory round trip.  One
 * at a time through the ring, such a step cost ~150 issue cycles for ~20 instructions of arithmetic (slot
 * bookkeeping, a scalar record request nobody reads, three LDS reads, the ring's DMA): the homogeneous family --
 * long road / sky stretches in which every split is a near-optimal candidate, 29 % of all pairs are such steps --
 * ran 25 % below the scene.  Here lane l fetches the four dwords a step reads of the record of vB - nwv (l & 15),
 * the next batch is requested before this one is walked, and step j takes its operands out of lane j with
 * v_readlane (a ROLLED loop: unrolled with DPP operands the sixteen steps of both types cost the walk's loop 30
 * spilled SGPRs, scene -7 %).  A candidate is two class differences, no instance term, no mean, no LUT value: a
 * fifth of the instructions of fast_step, same operand order (cost = dw * data + pw / h + sw * (f + nic)); the
 * object bound is sticky (the class minima only grow and the best cost cannot change any more).  Returns true when the type's
 * bound holds (the wave is done), else vB < lo.  lo = 0 includes the first segment: with the object type closed it
 * is a ground candidate like the others (record 0 is all zeros), for the lanes first_ok. */
template <bool SKY>
__device__ __forceinline__ bool gs_walk(const DevParams& P, const PruneValsF& pv, const RowRec& my,
                                        const RowRec* __restrict__ rcol, const float* s_rcp, int vTc, int& vB,
                                        const int lo, const int nwv, UnaryBestF& b, int& n_gs,
                                        const bool first_ok /* may this lane take the first segment (vB = 0)? */) {
    const int j16 = (int)(threadIdx.x & 15);
    /* SKY: Fsky (dword 18), Fnic (19), K (21); ground: Fg0 (0), Fg1 (1), Fnic (19), G (20) */
    auto fetch = [&](int v0, float& A, float& Bq, float& C, float& Dq) {
        const float* q = (const float*)(rcol + max(v0 - nwv * j16, lo)); /* (clamped: the loads are unconditional) */
        if (SKY) { A = q[18]; Bq = q[19]; C = q[21]; Dq = 0.0f; }
        else { A = q[0]; Bq = q[1]; C = q[19]; Dq = q[20]; }
    };
    float a0, a1, a2, a3, n0, n1, n2, n3;
    fetch(vB, a0, a1, a2, a3);
    while (vB >= lo) {
        fetch(vB - 16 * nwv, n0, n1, n2, n3);
        const int n_here = min(16, (vB - lo) / nwv + 1);
        bool closed = false;
        int jj = 0;
#pragma unroll 1
        for (; jj < n_here && !closed; jj++) {
            const int v = vB - jj * nwv;
            const float r = s_rcp[vTc + 1 - v];
            const float pwih = P.pw * r;
            const float q0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a0), jj));
            const float q1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a1), jj));
            const float q2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a2), jj));
            if (SKY) {
                const float f = my.Fsky - q0;
                const float nic = P.iw * (float)(my.Fnic - __float_as_int(q1));
                const float seg = f + nic;
                const float cost = P.dw * (my.K - q2) + pwih + P.sw * seg;
                take_if_le(b.s, b.vs, cost, v);
                const float lb = P.sw * seg - pv.E1s;
                closed = (__builtin_amdgcn_ballot_w64(lb > b.s) | pv.dead) == ~0ull;
            } else {
                const float q3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a3), jj));
                const float f = __builtin_fminf(my.Fg0 - q0, my.Fg1 - q1);
                const float nic = P.iw * (float)(my.Fnic - __float_as_int(q2));
                const float seg = f + nic;
                float cost = P.dw * (my.G - q3) + pwih + P.sw * seg;
                if (v == 0 && !first_ok) cost = IS_INF; /* :509: the first segment is ground only up to the horizon */
                take_if_le(b.g, b.vg, cost, v);
                const float lb = P.sw * seg - pv.E1g;
                closed = (__builtin_amdgcn_ballot_w64(lb > b.g) | pv.gdead) == ~0ull;
            }
        }
        n_gs += jj;
        if (closed) return true;
        vB -= n_here * nwv;
        a0 = n0; a1 = n1; a2 = n2; a3 = n3;
    }
    return false;
}

#ifdef IS_ABL_PHASES
__device__ unsigned long long g_fphase[8];
#define ISF_MARK_INIT() unsigned long long t_phase = __builtin_readcyclecounter()
#define ISF_MARK(k)                                                                       \
    do {                                                                                  \
        const unsigned long long now__ = __builtin_readcyclecounter();                    \
        if (threadIdx.x == 0) atomicAdd(&g_fphase[k], now__ - t_phase);                   \
        t_phase = now__;                                                                  \
    } while (0)
extern "C" void isk_debug_phases(unsigned long long* out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fphase), sizeof(g_fphase));
    if (reset) {
        unsigned long long z[8] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fphase), z, sizeof(z));
    }
}
#else
#define ISF_MARK_INIT()
#define ISF_MARK(k)
#endif

/* LDS-DMA of the lutT row and the record of vB into one ring slot: NVR + 1 VMEM instructions,
 * always issued (a uniform count for wait_vmcnt), no registers, no waiting. */
template <int NVR>
__device__ __forceinline__ void ring_prefetch(const float* __restrict__ lcol,
                                              const RowRec* __restrict__ rcol, int vB, int D,
                                              float* slot_row, float* slot_rec, int lane) {
    const float* row = lcol + (size_t)vB * D;
    const unsigned base = lds_addr(slot_row);
#pragma unroll
    for (int j = 0; j < NVR; j++) {
        const int f = min(lane + 64 * j, D - 1); /* lanes beyond the row re-read its last element */
        dma_dword(row + f, base + 256 * j);
    }
    if (lane < ISF_REC_F) /* 32 lanes x 4 B = the record; the instruction is issued by every wave */
        dma_dword((const float*)(rcol + vB) + lane, lds_addr(slot_rec));
}

/* The windowed ring slot: [IS_P1_WIN floats of lutT row vB from column lo][the 32 dwords of the record of vB] --
 * ONE LDS-DMA instruction, lanes 0-31 the row window, lanes 32-63 the record. */
/* (gbase / gstride: this lane's address at vB = 0 and its bytes per vB -- the lutT column IS_WIN_COL(win, lane) with
 * a row of D floats, or dword lane - 32 of the records of 128 bytes: three registers held through the walk, one
 * multiply-add per fill) */
struct WinLane {
    const char* gbase;
    unsigned gstride;
};
__device__ __forceinline__ WinLane win_lane(const float* __restrict__ lcol, const RowRec* __restrict__ rcol, int D, int win,
                                            int lane) {
    static_assert(IS_P1_WIN == 32 && ISF_REC_F == 32, "one 64-lane DMA = window + record");
    WinLane w;
    w.gbase = lane < 32 ? (const char*)(lcol + IS_WIN_COL(win, lane)) : (const char*)((const float*)rcol + (lane - 32));
    w.gstride = lane < 32 ? (unsigned)D * 4u : (unsigned)sizeof(RowRec);
    return w;
}
__device__ __forceinline__ void ring_prefetch_win(const WinLane& wl, int vB, float* slot) {
    dma_dword((const float*)(wl.gbase + (size_t)vB * wl.gstride), lds_addr(slot));
}

/* GEN: the slot is the record alone (32 lanes) */
__device__ __forceinline__ void ring_prefetch_rec(const RowRec* __restrict__ rcol, int vB, float* slot, int lane) {
    if (lane < ISF_REC_F) dma_dword((const float*)(rcol + vB) + lane, lds_addr(slot));
}

/* ====================================================================================== */
/* QDIAG: the diagonal block of a windowed tile in 16-lane quarters                            */
/* ====================================================================================== */
/* A wave of the windowed launch owns the vB = tile_lo + a of one residue class mod 4, a = a0, a0 - 4, ... >= 1 with
 * a0 = 63 - w: sixteen diagonal steps in which only the lanes vT >= vB hold a segment (32.5 of 64 on average), more
 * than half of a pruned walk's steps.  The rows 16 q .. 16 q + 15 of the tile (quarter q) need only the a <= 16 q + 15,
 * i.e. the list entries 4 (3 - q) .. 15: 16 + 12 + 8 + 4 = 40 quarter-steps.  A 16-lane row of the wave takes its vB
 * operands as DPP row_newbcast of ITS OWN two dwords (eval_segment_dpp), so the four quarters of a wave can each
 * walk their own vB: TEN steps with all four quarters busy instead of sixteen --
 *     lanes 48-63: rows of quarter 3, entries 0 .. 9            (all ten steps)
 *     lanes 32-47: rows of quarter 2, entries 4 .. 13
 *     lanes 16-31: rows of quarter 2, entries 14, 15 (steps 0, 1), then their own rows, entries 8 .. 15
 *     lanes  0-15: rows of quarter 3, entries 10 .. 15 (steps 0 .. 5), then their own rows, entries 12 .. 15
 * (entry 15 does not exist for w = 3: a dead quarter-step).  A quarter that changes rows hands its partial minima to
 * the lanes that own those rows (ds_bpermute + the merge rule of the waves: smaller cost, ties -> smaller vB) and
 * loads its own rows' record.  Everything a step reads is in LDS: both lutT windows (vB > tile_lo is a row of the
 * tile itself) and the records, of vB and of the lanes' rows alike (s_nat, below); candidates, operand order and the
 * `<=` update of a descending walk are fast_step's.  The tile that contains the horizon keeps the uniform steps (its
 * vB change from ground to sky candidates on the way).  Afterwards every lane holds its own row again and the wave
 * goes on below the tile. */
#ifndef ISF_QDIAG
#define ISF_QDIAG 1
#endif
/* Every record of the tile is staged in LDS (s_nat): the record of vB = tile_lo + a IS the record of row a - 1, so the
 * vB operands, the first record of every lane and the records the quarters 0 and 1 come back to are all LDS reads --
 * nothing in the diagonal phase waits for memory: 3.67 against 4.00 ms per 64 frames for staging only the rows
 * 0 .. 31 (the rest from global memory) and 4.23 ms for no staging at all: with the diagonal in quarters the kernel
 * executes 16 % fewer instructions, and what used to hide behind them -- a first record per wave, two reloads, a
 * cache line per step -- had become its critical path.
 * Row stride: 128 bytes, UNPADDED (a lane's whole-record read and the quarters' dword reads then meet in the same
 * banks: three record loads per wave, two dwords per step).  Padded to 144 bytes the records are 9.2 KB and the
 * workgroup 24.9 KB = six per CU at 1024 rows; unpadded, with a ring of two slots (ISF_RING), 22.8 KB = SEVEN.  With
 * the LUT units in the prepare launch that was a wash (3.65-3.71 against 3.70-3.73 ms), with the units inside this
 * launch -- they take workgroup slots -- the seventh slot is worth 1.2-2.4 % (11 030 -> 11 170-11 290 frames/s);
 * the records' 16-byte chunks XOR-swizzled by the row (conflict-free at 8 KB): 4.05 ms, the per-lane chunk addresses
 * cost 9 spilled VGPRs. */
#ifndef ISF_NAT_STRIDE
#define ISF_NAT_STRIDE 32
#endif
#define ISF_NAT_F (IS_TILE * ISF_NAT_STRIDE)
/* the row of the tile a lane works for first: the quarters 0 and 1 begin with rows of the quarters 3 and 2 */
__device__ __forceinline__ int qd_first_row(const int lane) {
    const int q = lane >> 4, l15 = lane & 15;
    return (q == 0) ? 48 + l15 : ((q == 1) ? 32 + l15 : lane);
}
template <bool HAS_INVALID, bool SKY>
__device__ __forceinline__ void diag_quarters(const DevParams& P, RowRec& my, UnaryBestF& b,
                                              const float* __restrict__ lcol,
                                              const float* s_tile, const float* s_rcp, const float* s_nat,
                                              const int tile_lo, const int w, const int win_lo, int& n_winmiss) {
    const int H = P.H, D = P.D;
    constexpr int DPW = IS_P1_WIN + 1;
    const int lane = (int)(threadIdx.x & 63), l15 = lane & 15, q = lane >> 4;
    const int a0 = IS_TILE - 1 - w;
    /* per lane, between two changes of rows: r the row of the tile it works for, hr = vTc + 1 - tile_lo, lim: the
     * candidate a holds a segment of the row iff 1 <= a <= lim (0 for a row beyond the image), trow its window row */
    int r, hr, a;
    unsigned lim;
    const float* trow;
    auto init_best = [&]() {
        b.g = b.o = b.s = IS_INF;
        b.vg = b.vs = -1;
        b.vo = 0;
    };
    /* dwords l15 and 16 + l15 of the record of vB = tile_lo + max(aa, 1) = the staged record of row max(aa, 1) - 1 */
    auto rec_dw = [&](int aa, float& r0, float& r1) {
        const float* pl = s_nat + (max(aa, 1) - 1) * ISF_NAT_STRIDE + l15;
        r0 = pl[0];
        r1 = pl[16];
    };
    auto take_rows = [&](const int row, const int entry) {
        r = row;
        const int vTc = min(tile_lo + r, H - 1);
        hr = vTc + 1 - tile_lo;
        lim = (tile_lo + r < H) ? (unsigned)(hr - 1) : 0u;
        trow = s_tile + r * DPW;
        a = a0 - 4 * entry;
        init_best();
    };
    /* the partial minima of the lanes `delta` below, which worked for the same rows, merged into the lanes `mine` */
    auto hand_over = [&](const int delta, const bool mine) {
        const int src = (mine ? lane - delta : lane) << 2;
        auto one = [&](float& c, int& vb) {
            const float c2 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, c)));
            const int vb2 = __builtin_amdgcn_ds_bpermute(src, vb);
            const bool take = mine && ((c2 < c) || (c2 == c && vb2 >= 0 && (vb < 0 || vb2 < vb)));
            c = take ? c2 : c;
            vb = take ? vb2 : vb;
        };
        one(b.g, b.vg);
        one(b.o, b.vo);
        one(b.s, b.vs);
    };
    take_rows(qd_first_row(lane), (q == 3) ? 0 : ((q == 2) ? 4 : ((q == 1) ? 14 : 10)));
    my = load_rec((const RowRec*)(s_nat + r * ISF_NAT_STRIDE));
    float R0, R1;
    rec_dw(a, R0, R1);
#pragma unroll 1
    for (int j = 0; j < 10; j++) {
        if (j == 2 || j == 6) { /* quarter 1 (j = 2) / quarter 0 (j = 6) goes back to its own rows */
            hand_over(j == 2 ? 16 : 48, q == (j == 2 ? 2 : 3));
            if (q == (j == 2 ? 1 : 0)) {
                take_rows(lane, j == 2 ? 8 : 12);
                my = load_rec((const RowRec*)(s_nat + lane * ISF_NAT_STRIDE)); /* this row's record, staged by the prologue */
                rec_dw(a, R0, R1);
            }
        }
        float N0, N1;
        rec_dw(a - 4, N0, N1); /* the next entry (a quarter that changes rows fetches its own after the change) */
        const int ac = max(a, 1);
        const bool live = (unsigned)(a - 1) < lim;
        const int hc = max(hr - ac, 1);
        const float rh = s_rcp[hc];
        constexpr int WANT = SKY ? IS_WANT_SKY : IS_WANT_GROUND;
        const SegTerms t = eval_segment_dpp<HAS_INVALID, WANT>(my, R0, R1, (float)hc, rh, D, P.iw, s_rcp);
        const int fo = IS_WIN_FIND(win_lo, t.fni);
        const bool inw = fo >= 0;
        const int foc = inw ? fo : 0;
        float vtv = trow[foc];
        float vbv = s_tile[(int)__umul24((unsigned)(ac - 1), (unsigned)DPW) + foc];
        const int vB = tile_lo + ac;
        const bool need = live && !inw;
        if (__builtin_amdgcn_ballot_w64(need) != 0ull) {
            if (need) {
                vtv = (lcol + (size_t)min(tile_lo + hr, H) * D)[(unsigned)t.fni];
                vbv = (lcol + (size_t)vB * D)[(unsigned)t.fni];
            }
            n_winmiss++;
        }
        const float od = vtv - vbv;
        const float pwih = P.pw * rh;
        const float cost_o = P.dw * od + pwih + P.sw * t.seg_o; /* left to right, fast_step */
        const bool uo = live && (cost_o <= b.o);
        b.o = uo ? cost_o : b.o;
        b.vo = uo ? vB : b.vo;
        if (SKY) {
            const float cost_s = P.dw * t.sd + pwih + P.sw * t.seg_s;
            const bool us = live && (cost_s <= b.s);
            b.s = us ? cost_s : b.s;
            b.vs = us ? vB : b.vs;
        } else {
            const float cost_g = P.dw * t.gd + pwih + P.sw * t.seg_g;
            const bool ug = live && (cost_g <= b.g);
            b.g = ug ? cost_g : b.g;
            b.vg = ug ? vB : b.vg;
        }
        a -= 4;
        R0 = N0;
        R1 = N1;
    }
}

/* PRE_DIAG: the instantiation that starts from the minima k_dp_unary_diag left in the tables (an
 * instantiation of its own: as a run-time flag the path cost the unpruned walk 1.8 %) */
/* ====================================================================================== */
/* LUTF: the LUT units of the prepare launch INSIDE the DP launch                            */
/* ====================================================================================== */
/* k_prepare_fused's LUT blocks are bound by their HBM writes (8.6 GB per 64 frames, 1.6 ms alone), this kernel by
 * VALU issue: side by side on the SAME CUs the two use different resources.  Two streams do not get there (a
 * second kernel only receives the CUs the first one leaves: measured, rounds 2 and 4) and a CU partition does not
 * either (a CU that runs prepare blocks cannot lend its VALU: round 5), so the LUT units become workgroups of
 * THIS launch, interleaved in block order with the DP workgroups: block IDs are dispatched in order (per XCD:
 * block b runs on XCD (b + k) mod 8 with ONE offset k for the whole launch -- the dispatcher's round-robin is not reset
 * between launches: measured in round 6, when the same units as a kernel of their own on a second stream landed one and
 * three XCDs away from their readers (profiles/r06_ab_lut_side_kernel.log); only "b and b + 8 share an XCD" is used),
 * so a DP workgroup that waits for its column's units -- a counter per column,
 * release / acquire -- only ever waits for workgroups that were dispatched before it.
 *   block order: [8 LEAD LUT blocks] then per super-group s of cpb = 4 / fn_blocks column groups: [8 LUT blocks of
 *   super-group s + LEAD] [cpb 8 ntl DP blocks of super-group s].  LUT block r of a super-group holds the units of
 *   its cpb columns with column mod 8 = r: they run on the XCD whose L2 the DP workgroups of those columns use.
 * The unit is object_lut_body (is_k_prepare.hip) without its software prefetch: 45 instead of 97 VGPRs, the kernel
 * keeps its 72-register budget, and the DP waves of the CU hide the unit's latency.
 * MEASURED (64 frames, `tools/run_variants.sh`, rocprofv3 averages): this launch 5.6 ms against 4.45 ms for the DP
 * alone -- of the 1.6 ms the LUT blocks take in the prepare launch 0.45 ms disappear; with the units EMPTY
 * (timing-only) 4.55 ms: the block interleaving and the waits cost nothing; with one store per block instead of
 * 32 (timing-only) 4.95 ms: it is the 8.6 GB write stream that slows the DP workgroups down (their prologues are
 * chains of memory round trips, and the round trips get longer), not the units' instructions -- a form of the
 * unit with scalar-loaded row offsets and a third of the VALU instructions ran no faster, nor did the prefetching
 * form at 6 or 5 waves per SIMD, a ring of 4 slots, or 8 / 128 / 512 super-groups of lead (8: 7.1 ms, the DP waits). */
#ifndef ISF_OCC_LUTF
#define ISF_OCC_LUTF 7
#endif
#ifndef ISF_OCC_LUTF_INV
#define ISF_OCC_LUTF_INV 6 /* with an invalid-disparity value: 80 VGPRs; at 7: 6 spilled VGPRs (see ISF_OCC_INV) */
#endif
#ifndef ISF_LUTF_MIN_COLS
#define ISF_LUTF_MIN_COLS 2048 /* columns per call from which the LUT units run inside the DP launch by default */
#endif
#ifndef ISF_LUTF_POLL_MAX
#define ISF_LUTF_POLL_MAX (1 << 14) /* polls of a DP workgroup for its column's units before it distrusts the hand-over */
#endif
#ifndef ISF_LUTF_LEAD
#define ISF_LUTF_LEAD 32 /* super-groups of LUT blocks dispatched ahead of the first DP block (8192 DP blocks at 1024 rows) */
#endif
/* the XCD (its L2) this wave runs on */
__device__ __forceinline__ int xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return (int)(x & 0xfu);
}
__device__ __forceinline__ void lut_unit_fused(const DevParams& P, const int colg, const int fn_block, const int lane,
                                               const float* __restrict__ joined, const float* __restrict__ cost_T,
                                               float* __restrict__ lutT, int* __restrict__ ready) {
    constexpr int LB = 32; /* LUT_BLOCK of k_object_lut */
    const int H = P.H, D = P.D;
    const int fn = fn_block * 64 + lane;
    const bool fn_ok = fn < D;
    const int fnc = fn_ok ? fn : D - 1;
    const float* dcol = joined + (size_t)colg * H;
    float* lcol = lutT + (size_t)colg * (H + 1) * D;
    if (fn_ok) lcol[fn] = 0.0f; /* arr[0] = 0, :283-285 */
    float add = 0.0f;
    /* (full blocks store unconditionally -- a lane beyond D holds lane D - 1's values and writes them to its address
     * again --: the same number of memory operations on every path, like object_lut_body) */
    auto block = [&](const int i, const bool full) {
        const int rl = i + (lane & (LB - 1));
        int dis_l = (rl < H) ? (int)dcol[rl] : 0; /* rows beyond the image use dis = 0, :244-247 */
        dis_l = min(max(dis_l, 0), D - 1);
        float c[LB];
#pragma unroll
        for (int l = 0; l < LB; l++) c[l] = cost_T[(size_t)__builtin_amdgcn_readlane(dis_l, l) * D + fnc];
        c[0] += add; /* :249-251 */
#pragma unroll
        for (int j = 1; j < LB; j <<= 1) {
#pragma unroll
            for (int l = LB - 1; l >= j; l--) c[l] += c[l - j];
        }
        float* dst = lcol + (size_t)(i + 1) * D + fnc;
        if (full) {
#ifdef ISF_ABL_LUTF_NOSTORE /* timing-only ablation: one row per block is stored (the network stays alive) */
            __builtin_nontemporal_store(c[LB - 1], dst + (size_t)(LB - 1) * D);
#else
#pragma unroll
            for (int l = 0; l < LB; l++) __builtin_nontemporal_store(c[l], dst + (size_t)l * D);
#endif
        } else if (fn_ok) {
#pragma unroll
            for (int l = 0; l < LB; l++)
                if (i + l < H) dst[(size_t)l * D] = c[l];
        }
        add = c[LB - 1]; /* :268-272 */
    };
#ifndef ISF_ABL_LUTF_EMPTY /* timing-only ablation: the units do nothing (the table of an earlier IS_LUT_FUSED=0 call is still there) */
    int i = 0;
#pragma unroll 1
    for (; i + LB <= H; i += LB) block(i, true);
    if (i < H) block(i, false);
#endif
    /* The unit's rows are visible before its count is: the stores have been acknowledged by the L2 (vmcnt(0)) that
     * the column's DP workgroups read through -- LUT block and DP workgroups of a column share their XCD (block ID mod
     * 8: OBSERVED placement, not a contract, so the unit publishes its XCC id and a reader that does not share it sets
     * DevParams::lutf_bad, which makes the launches behind this one do the call again) -- and the count is an L2 atomic.  (An agent-scope release fence is a write-back of the whole L2 on
     * gfx950, an agent-scope acquire an invalidation: at one per unit / workgroup the launch took 10.6 ms.) */
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    /* count + 256 x the XCC id of this unit (lut_fused == 2, tests: a wrong one): the readers check that they share it */
    if (lane == 0)
        __hip_atomic_fetch_add(ready + colg, 1 + (((xcc_id() + (P.lut_fused == 2 ? 1 : 0)) & 15) << 8), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
}

/* GEN (with WIN): lutT holds only its carry rows (DevParams::lut_carry); the tile and the vB-side rows are rebuilt */
template <bool HAS_INVALID, int NVR, bool PRE_DIAG = false, bool WIN = false, bool GEN = false, bool LUTF = false,
          bool REPAIR = false>
__global__ __launch_bounds__(ISF_THREADS, GEN ? ISF_OCC_GEN : (LUTF ? (HAS_INVALID ? ISF_OCC_LUTF_INV : ISF_OCC_LUTF) : (HAS_INVALID ? ISF_OCC_INV : ISF_OCC))) void k_dp_unary_fast(
    const DevParams P, int ncols, const RowRec* __restrict__ recs, const float* __restrict__ lutT,
    const float* __restrict__ rcp, const int* __restrict__ vhor_arr,
    const int* __restrict__ col_flags, const PruneRec* __restrict__ prune,
    float* __restrict__ cost_table, int32_t* __restrict__ index_table,
    unsigned long long* __restrict__ counters /* null, or the evaluation counters (is_device.h) */,
    const float* __restrict__ joined, const float* __restrict__ cost_T,
    int pre_diag /* k_dp_unary_diag has run: the tables hold the minima over the vB inside the tiles */,
    int tile0, int ntl /* this launch walks the tiles tile0 .. tile0 + ntl - 1 */,
    const int* __restrict__ run_if /* REPAIR: a word; the launch leaves at once while it is 0 */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(!REPAIR || (WIN && !GEN && !LUTF && !PRE_DIAG), "the repair launch is the plain windowed one");
    /* REPAIR (the launch behind a fused LUT + DP launch, DevParams::lutf_bad): a SMALL grid whose workgroups walk the
     * (column, tile) items of the plain launch one after the other -- 262144 workgroups that only read a word and
     * leave took 62 us per call. */
    if (REPAIR && __builtin_amdgcn_readfirstlane(*run_if) == 0) return;
    if (REPAIR && blockIdx.x == 0 && threadIdx.x == 0 && P.lutf_repairs != nullptr) /* a repaired call: the host's count */
        __hip_atomic_fetch_add(P.lutf_repairs, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const int H = P.H, D = P.D;
    const int DP = (WIN ? IS_P1_WIN : D) + 1; /* WIN: the tile holds an fn window of its rows (is_device.h) */
    const int nwv = (int)(blockDim.x >> 6);   /* waves of the workgroup: 8, or 4 for the windowed tiles (GEN: 4, the stride its row residues assume) */
    constexpr int K = ISF_RING;
    static_assert(!GEN || (WIN && !PRE_DIAG), "the rebuilt rows are window rows");
    constexpr int ROWF = GEN ? 0 : (WIN ? IS_P1_WIN : 64 * NVR); /* floats of a row slot (WIN: the fn window of the row; GEN: none, the rows are rebuilt) */
    constexpr int NV = WIN ? 1 : NVR + 1;            /* VMEM instructions per slot fill */
    constexpr int SLOT = ROWF + ISF_REC_F;     /* floats of a ring slot */
    float* s_rcp = (float*)smem;                        /* [H+1 -> x4]                       */
    float* s_tile = s_rcp + ((H + 1 + 3) & ~3);         /* [64][D+1] lutT rows tile_lo+1 ..   */
    /* (the tile's space also holds the merge area after the walk: at least ISF_MERGE_F floats) */
    const int merge_f = 2 * nwv * 3 * 64 + 2 * 3 * 64;
    const int tile_f = max((IS_TILE * DP + 3) & ~3, merge_f);
    float* s_ring = s_tile + tile_f;                    /* [8 waves][K][SLOT]                */
    float* s_cache = s_ring + (size_t)nwv * K * SLOT;   /* GEN: [waves][ISF_GEN_CACHE_F] rebuilt vB-side rows */
    float* s_nat = s_cache;                             /* QDIAG (never with GEN): [32][ISF_NAT_STRIDE] the records of the rows 0 .. 31 of the tile */
    float* s_zero = s_cache + (size_t)nwv * ISF_GEN_CACHE_F; /* GEN: [IS_P1_WIN] lutT row 0 */

    /* XCD-aware order: blocks b, b+8, ... share an XCD/L2; the tiles of a column stay on one XCD
     * (they fetch the same lutT rows) and the tallest tiles start first */
    const int nxcd = 8;
    int dpb = (int)blockIdx.x; /* index of this workgroup among the DP workgroups */
    if (LUTF) {
        static_assert(!LUTF || (WIN && !GEN && !PRE_DIAG), "the fused LUT units belong to the windowed launch of every tile");
        const int fnb = (D + 63) >> 6, cpb = 4 / fnb; /* (column, 64 fn) units and columns per LUT block: 4 waves */
        const int nSG = ((ncols + 7) / 8 + cpb - 1) / cpb, dp_per = cpb * 8 * ntl;
        const int lead = min(ISF_LUTF_LEAD, nSG);
        int b = (int)blockIdx.x, sg = -1;
        if (b < 8 * lead) {
            sg = b >> 3;
        } else {
            b -= 8 * lead;
            const int nfull = nSG - lead, per = 8 + dp_per;
            if (b < nfull * per) {
                const int s_ = b / per, o = b - s_ * per;
                if (o < 8) sg = s_ + lead;
                else dpb = s_ * dp_per + (o - 8);
            } else {
                dpb = nfull * dp_per + (b - nfull * per);
            }
        }
        if (sg >= 0) { /* a LUT block: wave w takes unit w of the columns (sg cpb + k) 8 + r, r = this block's XCD */
            const int wv = (int)(threadIdx.x >> 6), r = (int)blockIdx.x & 7;
            const int col = ((sg * cpb + wv / fnb) * 8) + r;
            const unsigned long long t0 = counters != nullptr ? __builtin_readcyclecounter() : 0ull;
            if (col < ncols) lut_unit_fused(P, col, wv % fnb, (int)(threadIdx.x & 63), joined, cost_T, const_cast<float*>(lutT), P.lut_ready);
            if (counters != nullptr && (threadIdx.x & 63) == 0) /* (measurements only: the unit's life in shader clocks) */
                atomicAdd(counters + IS_CNT_LUTF_UNIT_CYCLES, __builtin_readcyclecounter() - t0);
            return;
        }
    }
    if (REPAIR) dpb = (int)blockIdx.x;
next_item: /* (REPAIR: the next (column, tile) item of this workgroup) */
    {
    const int xcd = dpb % nxcd, q = dpb / nxcd;
    const int tile = __builtin_amdgcn_readfirstlane(tile0 + ntl - 1 - q % ntl);
    const int colg = __builtin_amdgcn_readfirstlane((q / ntl) * nxcd + xcd);
    if (colg >= ncols) goto item_done;
    if (__builtin_amdgcn_readfirstlane(col_flags[colg]) != 0) goto item_done; /* generic column: k_dp_unary */
    const int vhor = __builtin_amdgcn_readfirstlane(vhor_arr[colg / P.C]);
    /* WIN: first lutT column of the fn window of this (column, tile) (k_prepare writes it; requested with the flags) */
    const int win_lo = WIN ? __builtin_amdgcn_readfirstlane(P.win_lo[(size_t)colg * P.ntiles + tile]) : 0;
    /* (pairs with a generic-encoding column are skipped by the diagonal kernel) */
    const bool pre = PRE_DIAG && pre_diag != 0 && (colg | 1) < ncols &&
                     (__builtin_amdgcn_readfirstlane(col_flags[colg ^ 1]) == 0);

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (LUTF) {
        /* this column's LUT units: workgroups of this launch with smaller block IDs, i.e. dispatched already (they
         * wait for nothing) and, block ID mod 8 being equal, on this workgroup's XCD.  Both are what the dispatcher is
         * OBSERVED to do, neither is a contract: the poll is bounded, the units publish their XCC id, and a workgroup
         * that runs out of polls or finds another id than its own sets lutf_bad -- the repair launches behind this
         * one then redo the call from the table of the ordinary LUT kernel (what this workgroup writes is overwritten). */
        if (tid == 0) {
            const int need = (D + 63) >> 6;
            int spins = 0, seen;
            /* bounded: ISF_LUTF_POLL_MAX polls of about 1.5 us (s_sleep 32 = 2048 clocks + the two loads), i.e. some tens
             * of milliseconds, and a workgroup leaves at once when another one has distrusted the hand-over already --
             * the call is redone anyway, and 262144 workgroups must not each wait out their bound */
            while (((seen = __hip_atomic_load(P.lut_ready + colg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 255) < need &&
                   ++spins < ISF_LUTF_POLL_MAX &&
                   __hip_atomic_load(P.lutf_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                __builtin_amdgcn_s_sleep(32);
            if ((seen & 255) < need || (seen >> 8) != need * xcc_id())
                __hip_atomic_fetch_or(P.lutf_bad, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (counters != nullptr && spins > 0) atomicAdd(counters + IS_CNT_LUTF_SPINS, (unsigned long long)spins);
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    const RowRec* rcol = recs + (size_t)colg * (H + 1);
    const float* lcol = lutT + (size_t)colg * (H + 1) * D;
    const int tile_lo = tile * IS_TILE;
    const int vT = tile_lo + lane;
    const int vTc = min(vT, H - 1);
    const bool row_ok = vT < H;
    const int vB_end = min(tile_lo + IS_TILE - 1, H - 1);
    float* my_ring = s_ring + (size_t)w * K * SLOT;
    ISF_MARK_INIT();

    /* ---- prologue: the wave's first K slots are requested first, then the tile's lutT rows,
     * the 1/h table and this lane's record.  The wave's steps: vB_top, vB_top - 8, ... >= 0
     * (H >= 8 = the number of waves, so vB_top >= 0). */
    /* QDIAG: the diagonal block in quarters (diag_quarters), the walk then starts below the tile */
    const bool qd = ISF_QDIAG && WIN && !GEN && !pre && nwv == 4 && (tile_lo >= vhor || tile_lo + IS_TILE - 1 <= vhor);
#ifdef ISF_ABL_NODIAG /* timing-only ablation (wrong results): the walk without its diagonal block */
    const int vB_top = tile_lo - w;
#else
    const int vB_top = ((pre || qd) ? tile_lo : vB_end) - w; /* (pre, qd: may be negative = no step for this wave) */
#endif
    const WinLane wlane = win_lane(lcol, rcol, D, win_lo, lane);
#pragma unroll
    for (int i = 0; i < K; i++)
    {
        const int vq = max(vB_top - nwv * i, 0);
        if (GEN) ring_prefetch_rec(rcol, vq, my_ring + i * SLOT, lane);
        else if (WIN) ring_prefetch_win(wlane, vq, my_ring + i * SLOT);
        else ring_prefetch<NVR>(lcol, rcol, vq, D, my_ring + i * SLOT, my_ring + i * SLOT + ROWF, lane);
    }
    ISF_MARK(4); /* (debug build: ring requests issued) */
    /* (qd: diag_quarters loads the record of the row this lane's quarter works for first -- requested here, in front
     * of the tile staging, it costs the kernel 43 spilled VGPRs --; the quarters 0 and 1 take their own rows'
     * records out of s_nat when they come back to them) */
    RowRec my;
    if (!qd) my = load_rec(rcol + vTc + 1);
    if (WIN && !GEN && qd) { /* 32 records x 8 float4 = one per thread of the 4-wave workgroup */
        float4 x[IS_TILE / 32];
#pragma unroll
        for (int k = 0; k < IS_TILE / 32; k++) {
            const int row = (tid >> 3) + 32 * k, ch = tid & 7;
            x[k] = reinterpret_cast<const float4*>(rcol + min(tile_lo + row, H - 1) + 1)[ch];
        }
#pragma unroll
        for (int k = 0; k < IS_TILE / 32; k++) {
            const int row = (tid >> 3) + 32 * k, ch = tid & 7;
            *reinterpret_cast<float4*>(s_nat + row * ISF_NAT_STRIDE + 4 * ch) = x[k];
        }
    }
    ISF_MARK(5); /* (debug build: record requested) */
    float* my_cache = s_cache + (size_t)w * ISF_GEN_CACHE_F;
    if (GEN) {
        /* wave 0 rebuilds the tile (its cache is the scratch of the row offsets), the others stage the 1/h table */
        if (w == 0) gen_window_tile(s_tile, (int*)my_cache, lcol, joined + (size_t)colg * H, cost_T, tile_lo, win_lo, H, D);
        else stage_rcp(s_rcp, rcp, H, tid - 64, (int)blockDim.x - 64);
        if (tid < IS_P1_WIN) s_zero[tid] = 0.0f; /* lutT[0][*] = 0, :283-285 */
    } else if (WIN) stage_window_and_rcp(s_tile, s_rcp, lcol, rcp, tile_lo, H, D, win_lo, tid, (int)blockDim.x);
    else stage_tile_and_rcp(s_tile, s_rcp, lcol, rcp, tile_lo, H, D, tid, (int)blockDim.x);
    int n_winmiss = 0;
    FastWin fwin;
    fwin.lo = win_lo;
    fwin.gcol = lcol;
    fwin.misses = &n_winmiss;
    fwin.dcol = joined + (size_t)colg * H;
    fwin.cost_T = cost_T;
    fwin.vT1 = min(vT + 1, H);
    ISF_MARK(6); /* (debug build: tile + 1/h table staged; mark 0 then = the barrier) */

    PruneValsF pv;
    {
        cprune_t pq = (cprune_t)(prune + colg);
        pv.E1o = pq->E1o; pv.E1g = pq->E1g; pv.E1s = pq->E1s; pv.E2 = 3.0f * pq->E2;
        pv.dead = ~__builtin_amdgcn_ballot_w64(row_ok);
        pv.gdead = pv.dead; /* (completed below, once this lane's record is here) */
    }
    fwin.E1o = pv.E1o;
    UnaryBestF b;
    b.g = b.o = b.s = IS_INF;
    b.vg = b.vs = -1;
    b.vo = 0; /* index_table[vT*3+OBJECT] = OBJECT at vB = 0, :592 */
    if (pre && row_ok) { /* every wave starts from the diagonal block's minima: the bounds bite at once */
        const size_t o = ((size_t)colg * H + vT) * 3;
        b.g = cost_table[o + 0]; b.o = cost_table[o + 1]; b.s = cost_table[o + 2];
        b.vg = index_table[o + 0]; b.vo = index_table[o + 1]; b.vs = index_table[o + 2];
    }
    const float* my_tile = s_tile + lane * DP;
    const bool nog = IS_SKIP_GROUND_ABOVE_HORIZON && tile_lo >= vhor;
    __syncthreads(); /* the tile and the 1/h table: the only data the waves share */
    ISF_MARK(0);
    if (WIN && !GEN && qd) {
        if (tile_lo >= vhor) diag_quarters<HAS_INVALID, true>(P, my, b, lcol, s_tile, s_rcp, s_nat, tile_lo, w, win_lo, n_winmiss);
        else diag_quarters<HAS_INVALID, false>(P, my, b, lcol, s_tile, s_rcp, s_nat, tile_lo, w, win_lo, n_winmiss);
    }
    ISF_MARK(7); /* (debug build: the diagonal quarters) */
    pv.gdead = pv.dead | __builtin_amdgcn_ballot_w64(my.G == IS_INF);

    /* ---- the wave's walk, vB downwards; slot i % K holds step i */
    int slot = 0;
    int gen_k = -100; /* GEN: the cache holds the rows of this wave's residue of the blocks gen_k and gen_k - 1 */
    auto gen_rebuild = [&](int kb, int rho) { /* (one residue per wave: vB steps by the 4 waves of the workgroup) */
        const float* dcol = joined + (size_t)colg * H;
        switch (rho) {
        case 0: gen_rebuild_rows<0>(my_cache, lcol, dcol, cost_T, kb, win_lo, H, D); break;
        case 1: gen_rebuild_rows<1>(my_cache, lcol, dcol, cost_T, kb, win_lo, H, D); break;
        case 2: gen_rebuild_rows<2>(my_cache, lcol, dcol, cost_T, kb, win_lo, H, D); break;
        default: gen_rebuild_rows<3>(my_cache, lcol, dcol, cost_T, kb, win_lo, H, D); break;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* the cache rows are written before they are read */
    };
    bool o_closed = false;
    int vB_gs = -1; /* first vB of the batched ground / sky-only walk (gs_walk), -1: none */
    int n_full = 0, n_gs = 0; /* steps below the diagonal block (wave-uniform: SALU only) */
    isk_f16v S; /* the class prefixes of the record of vB as scalars, requested one step ahead */
#if ISF_SREC
    srec_request(S, rcol + max(vB_top, 0));
#endif
    for (int vB = vB_top; vB >= 0; vB -= nwv) {
        wait_vmcnt<NV * (K - 1)>(); /* this step's slot has landed; K - 1 prefetches in flight */
#if ISF_SREC
        srec_arrived(S);
#endif
        float* s_row = my_ring + slot * SLOT;
        const float* rb = s_row + ROWF; /* the record of vB in the ring slot */
        const float* lrow = s_row;
        const bool diag = vB > tile_lo;
        if (GEN && !(ISF_GS_STEPS && o_closed && vB != 0)) { /* (the ground / sky-only walk reads no table entry) */
            if (diag) {
                lrow = s_tile + (vB - tile_lo - 1) * DP; /* a row of the tile itself */
            } else if (vB == 0) {
                lrow = s_zero;
            } else {
                const int kb = (vB - 1) >> 5; /* row vB = output (vB - 1) mod 32 of block kb */
                if (kb != gen_k && kb != gen_k - 1) {
                    gen_rebuild(kb, (vB - 1) & 3);
                    gen_k = kb;
                }
                lrow = my_cache + (gen_k - kb) * ISF_GEN_HALF_F + (((vB - 1) & 31) >> 2) * IS_P1_WIN;
            }
        }
        bool done = false;
        int ok = 0;
        if (ISF_GS_STEPS && o_closed && vB != 0) { /* (closed in a full step: the diagonal is over) */
            vB_gs = vB; /* the rest of the walk holds ground / sky candidates only: gs_walk, behind the loop */
            break;
        } else {
        n_full += diag ? 0 : 1;
        if (vB == 0) { /* first segment (:481-594): ground + object */
            if (diag)
                fast_step<HAS_INVALID, false, true, true, false, WIN, GEN>(P, my, rb, lrow, my_tile, s_rcp, vT, vTc, vhor,
                                                                      0, row_ok, b, S, fwin);
            else
                fast_step<HAS_INVALID, false, false, true, false, WIN, GEN>(P, my, rb, lrow, my_tile, s_rcp, vT, vTc,
                                                                       vhor, 0, row_ok, b, S, fwin);
        } else if (vB > vhor) { /* vB - 1 >= vhor: sky + object (:729) */
            if (diag) {
                fast_step<HAS_INVALID, true, true, false, false, WIN, GEN>(P, my, rb, lrow, my_tile, s_rcp, vT, vTc, vhor,
                                                                      vB, row_ok, b, S, fwin);
            } else {
                const SegTerms t = fast_step<HAS_INVALID, true, false, false, false, WIN, GEN>(
                    P, my, rb, lrow, my_tile, s_rcp, vT, vTc, vhor, vB, row_ok, b, S, fwin);
                if (IS_PRUNE) ok = fast_bounds<true, false>(P, pv, t, b);
            }
        } else { /* ground + object (:687) */
            if (diag) {
                fast_step<HAS_INVALID, false, true, false, false, WIN, GEN>(P, my, rb, lrow, my_tile, s_rcp, vT, vTc, vhor,
                                                                       vB, row_ok, b, S, fwin);
            } else if (nog) {
                const SegTerms t = fast_step<HAS_INVALID, false, false, false, true, WIN, GEN>(
                    P, my, rb, lrow, my_tile, s_rcp, vT, vTc, vhor, vB, row_ok, b, S, fwin);
                if (IS_PRUNE) ok = fast_bounds<false, true>(P, pv, t, b);
            } else {
                const SegTerms t = fast_step<HAS_INVALID, false, false, false, false, WIN, GEN>(
                    P, my, rb, lrow, my_tile, s_rcp, vT, vTc, vhor, vB, row_ok, b, S, fwin);
                if (IS_PRUNE) ok = fast_bounds<false, false>(P, pv, t, b);
            }
        }
        done = ok == 3;
        o_closed = (ok & 1) != 0; /* sticky; the ground / sky bound is re-tested in its own steps
                                   * (it does not carry over from the sky range to the ground range) */
        }
        if (done) break; /* nothing below can win any more */

        /* refill the slot just consumed (its reads have returned: their values were used) */
        asm volatile("" ::: "memory");
        if (GEN) ring_prefetch_rec(rcol, max(vB - nwv * K, 0), s_row, lane);
        else if (WIN) ring_prefetch_win(wlane, max(vB - nwv * K, 0), s_row);
        else ring_prefetch<NVR>(lcol, rcol, max(vB - nwv * K, 0), D, s_row, s_row + ROWF, lane);
#if ISF_SREC
        srec_request_next(S, rcol + max(vB - nwv, 0)); /* (in-out: the loop-carried value keeps its registers, no copy while the request is in flight) */
#endif
        slot = (slot + 1 == K) ? 0 : slot + 1;
    }
#if ISF_SREC
    srec_arrived(S); /* a walk that ran to vB = 0 leaves a request in flight: it must not land in
                      * registers that hold something else by then */
#endif
    if (vB_gs >= 1) {
        /* Sixteen candidates per round trip instead of one per ring slot (placed behind the loop: inside it its eight
         * operand registers cost the walk a spilled pointer, reloaded in every step).  A sky range below the tile
         * means that the whole tile lies above the horizon: once it is exhausted only +inf ground candidates and the
         * closed first-segment object candidate are left.  The ground range runs down to vB = 0: with the object
         * type closed, the first segment (:481-594) is its ground candidate alone -- the same expression on the
         * all-zero record 0, for the lanes vT <= vhor. */
        int v = vB_gs;
        if (v > vhor) (void)gs_walk<true>(P, pv, my, rcol, s_rcp, vTc, v, max(vhor + 1, 1), nwv, b, n_gs, true);
        else if (!nog) (void)gs_walk<false>(P, pv, my, rcol, s_rcp, vTc, v, 0, nwv, b, n_gs, vT <= vhor);
    }
    ISF_MARK(1);
    if (counters != nullptr && lane == 0) {
        atomicAdd(counters + IS_CNT_UNARY_FULL, (unsigned long long)n_full);
        atomicAdd(counters + IS_CNT_UNARY_GS, (unsigned long long)n_gs);
        if (WIN) atomicAdd(counters + IS_CNT_UNARY_WINMISS, (unsigned long long)n_winmiss);
    }

    /* ---- merge the waves' partial minima: min cost, ties -> smallest vB.  After the barrier no
     * wave reads the lutT tile any more, so the merge runs in ITS space; the rings may still
     * receive prefetched slots (a wave that left early has up to K fills in flight), and nothing
     * has to wait for them until the very end: the drain hides behind the barrier and the merge
     * (it used to sit in front of them: an HBM latency per workgroup). */
    __syncthreads();
    ISF_MARK(2);
    float* m_cost = s_tile;                              /* [8][3][64] */
    int* m_vb = (int*)(m_cost + nwv * 3 * 64);           /* [8][3][64] */
    float* f_cost = m_cost + 2 * nwv * 3 * 64;           /* [3][64] final values */
    int* f_vb = (int*)(f_cost + 3 * 64);                 /* [3][64] */
    m_cost[(w * 3 + 0) * 64 + lane] = b.g; m_vb[(w * 3 + 0) * 64 + lane] = b.vg;
    m_cost[(w * 3 + 1) * 64 + lane] = b.o; m_vb[(w * 3 + 1) * 64 + lane] = b.vo;
    m_cost[(w * 3 + 2) * 64 + lane] = b.s; m_vb[(w * 3 + 2) * 64 + lane] = b.vs;
    __syncthreads();
    if (tid < 3 * 64) {
        const int type = tid >> 6;
        float c = m_cost[(0 * 3 + type) * 64 + lane];
        int vb = m_vb[(0 * 3 + type) * 64 + lane];
        for (int ww = 1; ww < nwv; ww++) {
            const float c2 = m_cost[(ww * 3 + type) * 64 + lane];
            const int vb2 = m_vb[(ww * 3 + type) * 64 + lane];
            const bool take = (c2 < c) || (c2 == c && vb2 >= 0 && (vb < 0 || vb2 < vb));
            if (take) { c = c2; vb = vb2; }
        }
        /* a row without a finite candidate keeps the initial index (the descending walk records
         * +inf candidates, the reference's strict < never does; :592 for the object type) */
        if (!(c < IS_INF)) vb = (type == IS_OBJECT) ? 0 : -1;
        f_cost[type * 64 + lane] = c;
        f_vb[type * 64 + lane] = vb;
    }
    __syncthreads();
    if (w == 0 && row_ok) {
        const size_t o = ((size_t)colg * H + vT) * 3;
        float* cd = cost_table + o;
        int32_t* id = index_table + o;
        cd[0] = f_cost[0 * 64 + lane]; cd[1] = f_cost[1 * 64 + lane]; cd[2] = f_cost[2 * 64 + lane];
        id[0] = f_vb[0 * 64 + lane]; id[1] = f_vb[1 * 64 + lane]; id[2] = f_vb[2 * 64 + lane];
    }
    wait_vmcnt<0>(); /* no LDS-DMA may land after the workgroup has gone (its LDS is reassigned) */
    ISF_MARK(3);
    }
item_done:
    if (REPAIR) {
        __syncthreads(); /* (the item's LDS is free) */
        dpb += (int)gridDim.x;
        if (dpb < ((ncols + nxcd - 1) / nxcd) * nxcd * ntl) goto next_item;
    }
}

/* ====================================================================================== */
/* The diagonal blocks as a kernel of their own (IS_UNARY_DIAG)                              */
/* ====================================================================================== */
/* In k_dp_unary_fast a lane owns one row vT of a 64-row tile, so the 63 steps whose vB lies INSIDE
 * the tile run with the lanes vT < vB dead: 32.5 of 64 live on average, and on pruned walks these
 * steps are more than half of the kernel's work.  Here ONE wave takes the
 * diagonal blocks of a column PAIR (the decomposition of k_pw_phase2x, without its serial chain):
 * column X in lanes 0-31, column Y in lanes 32-63, and per column
 *     rows 32-63 against vB = tile_lo + 63 ... tile_lo + 1   (upper triangle, then the square),
 *     rows  0-31 against vB = tile_lo + 31 ... tile_lo + 1   (lower triangle):
 * 94 wave-steps for two columns instead of 2 x 63.  The vB record is a DPP operand (two dwords per
 * lane, fetched two steps ahead from the column's own record: a 16-lane row only ever reads its own
 * column), the lutT values of both ends come from the fn window of the tile (pw_phase2_body: a
 * segment inside the tile has its mean between the tile's smallest and largest disparity; a lane
 * outside the window reads global memory).  Candidates, operand order and the `<=` update of a
 * descending walk are those of fast_step.  The result -- the minima over the vB of the tile -- goes
 * into the rows of cost_table / index_table as PRELIMINARY values: k_dp_unary_fast (pre_diag) starts
 * every wave of the tile from them and walks vB <= tile_lo only.  Pairs with a generic-encoding
 * column are left alone (their FAST column walks its diagonal itself). */
#define ISD_WMAX 16
#define ISD_WS (ISD_WMAX + 1) /* odd row stride: lanes reading one window column of 32 rows hit 32 banks */
#define ISD_ROWS (IS_TILE + 1)
#define ISD_WF (ISD_ROWS * ISD_WS)
#ifndef ISD_OCC
#define ISD_OCC 5
#endif
__device__ __forceinline__ float isd_half_min(float x) {
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) x = __builtin_fminf(x, __shfl_xor(x, m, 64));
    return x;
}
__device__ __forceinline__ float isd_half_max(float x) {
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) x = __builtin_fmaxf(x, __shfl_xor(x, m, 64));
    return x;
}

template <bool HAS_INVALID>
__global__ __launch_bounds__(64, ISD_OCC) void k_dp_unary_diag(
    const DevParams P, int ncols, const RowRec* __restrict__ recs, const float* __restrict__ lutT,
    const float* __restrict__ joined, const float* __restrict__ rcp, const int* __restrict__ vhor_arr,
    const int* __restrict__ col_flags, float* __restrict__ cost_table, int32_t* __restrict__ index_table) {
    __shared__ float s_rcp[(IS_TILE + 1 + 3) & ~3];
    __shared__ float s_win[2 * ISD_WF];
    const int H = P.H, D = P.D;
    const int lane = threadIdx.x, li = lane & 31, l15 = lane & 15, half = lane >> 5;
    /* consecutive workgroups: the tiles of one column pair (neighbouring lutT rows, one XCD's L2 by
     * and large), tallest first is irrelevant here -- every workgroup does the same work */
    const int pair = (int)(blockIdx.x / (unsigned)P.ntiles), tile = (int)(blockIdx.x % (unsigned)P.ntiles);
    const int col0 = pair * 2;
    if (col0 + 1 >= ncols) return;
    if ((__builtin_amdgcn_readfirstlane(col_flags[col0]) | __builtin_amdgcn_readfirstlane(col_flags[col0 + 1])) != 0)
        return;
    const int vhor = __builtin_amdgcn_readfirstlane(vhor_arr[col0 / P.C]);
    const int tile_lo = tile * IS_TILE;
    const int colg = col0 + half;
    const RowRec* rcol = recs + (size_t)colg * (H + 1);
    const float* lcol = lutT + (size_t)colg * (H + 1) * D;
    float* my_winbase = s_win + half * ISD_WF;

    /* ---- the fn window [lo, lo + W) of this column's tile rows */
    int lo, W;
    {
        float dmin = IS_INF, dmax = -IS_INF;
        for (int k = 0; k < 2; k++) {
            const int v = tile_lo + li + 32 * k;
            const float d = joined[(size_t)colg * H + min(v, H - 1)];
            const bool ok = (v < H) && !(HAS_INVALID && d == P.invalid);
            dmin = __builtin_fminf(dmin, ok ? d : IS_INF);
            dmax = __builtin_fmaxf(dmax, ok ? d : -IS_INF);
        }
        dmin = isd_half_min(dmin);
        dmax = isd_half_max(dmax);
        int l = (int)__builtin_fminf(__builtin_fmaxf(dmin, 1.0f), (float)D) - 1;
        l = min(max(l, 0), D - 1);
        int hh = (int)__builtin_fminf(__builtin_fmaxf(dmax, 0.0f), (float)(D - 1)) + 1;
        hh = min(max(hh, l), min(D - 1, l + ISD_WMAX - 1));
        lo = l;
        W = hh - l + 1;
    }
    { /* window rows tile_lo .. tile_lo + 64: 32 lanes, 16 columns at most, two rows per sweep */
        constexpr int NL = (ISD_ROWS * ISD_WMAX + 31) / 32;
        const int f = li & (ISD_WMAX - 1), j0 = li >> 4;
        float tmp[NL];
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int j = j0 + 2 * k;
            tmp[k] = (j < ISD_ROWS && f < W) ? lcol[(size_t)min(tile_lo + j, H) * D + lo + f] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int j = j0 + 2 * k;
            if (j < ISD_ROWS && f < W) my_winbase[j * ISD_WS + f] = tmp[k];
        }
    }
    for (int i = lane; i <= IS_TILE; i += 64) s_rcp[i] = rcp[min(i, H)];
    __syncthreads();

    /* one phase: the rows base + li of the column against vB = tile_lo + a_hi ... tile_lo + 1 */
    auto phase = [&](const int base, const int a_hi) {
        const int r = base + li;            /* row of the tile */
        const int vT = tile_lo + r, vTc = min(vT, H - 1);
        const bool row_ok = vT < H;
        const RowRec my = load_rec(rcol + vTc + 1);
        const float* my_win = my_winbase + (r + 1) * ISD_WS; /* lutT row vT + 1 */
        UnaryBestF b;
        b.g = b.o = b.s = IS_INF;
        b.vg = b.vs = -1;
        b.vo = 0;
        /* records of vB two steps ahead (rows beyond H are never live: clamped for the address only) */
        auto rec_dw = [&](int a, float& r0, float& r1) {
            const float* q = (const float*)(rcol + min(max(tile_lo + a, 0), H));
            r0 = q[l15];
            r1 = q[16 + l15];
        };
        float c0, c1, n0, n1;
        rec_dw(a_hi, c0, c1);
        rec_dw(a_hi - 1, n0, n1);
        for (int a = a_hi; a >= 1; a--) {
            const int vB = tile_lo + a;
            const float R0 = c0, R1 = c1;
            c0 = n0; c1 = n1;
            rec_dw(a - 2, n0, n1);
            const int h = vTc + 1 - vB;
            const bool live = (h > 0) && row_ok;
            const int hc = max(h, 1);
            const float rh = s_rcp[min(hc, IS_TILE)];
            const bool sky = vB > vhor; /* vB - 1 >= vhor: sky + object (:729), else ground + object (:687) */
            SegTerms t;
            if (sky) t = eval_segment_dpp<HAS_INVALID, IS_WANT_SKY>(my, R0, R1, (float)hc, rh, D, P.iw, s_rcp);
            else t = eval_segment_dpp<HAS_INVALID, IS_WANT_GROUND>(my, R0, R1, (float)hc, rh, D, P.iw, s_rcp);
            const int fo = t.fni - lo;
            const bool inwin = (unsigned)fo < (unsigned)W;
            const int foc = inwin ? fo : 0;
            float od = my_win[foc] - my_winbase[a * ISD_WS + foc];
            if (__builtin_amdgcn_ballot_w64(live && !inwin) != 0ull) { /* outside the window: rare */
                const float og = (lcol + (size_t)(vTc + 1) * D)[(unsigned)t.fni] -
                                 (lcol + (size_t)min(vB, H) * D)[(unsigned)t.fni];
                od = inwin ? od : og;
            }
            const float pwih = P.pw * rh;
            /* cost = dw*data + pw*(1/h) + sw*seg, left to right (fast_step) */
            const float cost_o = P.dw * od + pwih + P.sw * t.seg_o;
            const bool uo = live && (cost_o <= b.o);
            b.o = uo ? cost_o : b.o;
            b.vo = uo ? vB : b.vo;
            if (sky) {
                const float cost_s = P.dw * t.sd + pwih + P.sw * t.seg_s;
                const bool us = live && (cost_s <= b.s);
                b.s = us ? cost_s : b.s;
                b.vs = us ? vB : b.vs;
            } else {
                const float cost_g = P.dw * t.gd + pwih + P.sw * t.seg_g;
                const bool ug = live && (cost_g <= b.g);
                b.g = ug ? cost_g : b.g;
                b.vg = ug ? vB : b.vg;
            }
        }
        if (row_ok) {
            const size_t o = ((size_t)colg * H + vT) * 3;
            cost_table[o + 0] = b.g; cost_table[o + 1] = b.o; cost_table[o + 2] = b.s;
            index_table[o + 0] = b.vg; index_table[o + 1] = b.vo; index_table[o + 2] = b.vs;
        }
    };
    phase(32, 63);
    phase(0, 31);
}

extern "C" {

hipError_t isk_launch_lut_repair(const DevParams* P, int ncols, const float* joined, const float* cost_T, float* lutT,
                                 hipStream_t stream); /* is_k_prepare.hip */

/* NVR = 64-lane loads per lutT row; 0 = the shape cannot use this kernel */
static int isf_nvr(const DevParams* P) {
    if (P->D <= 128) return 2;
    if (P->D <= 256) return 4;
    return 0;
}

static size_t isf_lds_bytes(const DevParams* P, int nvr, int nwaves, bool windowed, bool gen = false) {
    const size_t DP = (size_t)(windowed ? IS_P1_WIN : P->D) + 1;
    const size_t rcp = ((size_t)P->H + 1 + 3) & ~(size_t)3;
    size_t tile = ((size_t)IS_TILE * DP + 3) & ~(size_t)3;
    size_t ring = (size_t)nwaves * ISF_RING * ((gen ? 0 : (windowed ? (size_t)IS_P1_WIN : 64 * (size_t)nvr)) + ISF_REC_F);
    if (gen) ring += (size_t)nwaves * ISF_GEN_CACHE_F + IS_P1_WIN; /* the rebuilt rows + the zero row */
    else if (windowed && ISF_QDIAG) ring += ISF_NAT_F;              /* the records of the rows 0 .. 31 (diag_quarters) */
    const size_t merge = (size_t)nwaves * 3 * 64 * 2 + 2 * 3 * 64; /* lives in the tile's space */
    if (tile < merge) tile = merge;
    return sizeof(float) * (rcp + tile + ring) + 16;
}
size_t isk_unary_fast_lds_bytes(const DevParams* P, int nvr) { return isf_lds_bytes(P, nvr, ISF_WAVES, false); }

/* nonzero (the row-load count) when the shape can use the kernel: a workgroup must fit the CU's
 * 160 KiB of LDS */
int isk_unary_fast_chunk_rows(const DevParams* P) {
    const int nvr = isf_nvr(P);
    if (nvr == 0 || P->H < ISF_WAVES) return 0;
    return isk_unary_fast_lds_bytes(P, nvr) <= 160 * 1024 ? nvr : 0;
}

hipError_t isk_set_lds_unary_fast(const DevParams* P) {
    const int nvr = isk_unary_fast_chunk_rows(P);
    if (nvr == 0) return hipSuccess;
    size_t bb = isk_unary_fast_lds_bytes(P, nvr);
    for (int gen = 0; gen < 2; gen++) { /* (the windowed forms are smaller at every shape in use; not relied upon) */
        const size_t w = isf_lds_bytes(P, nvr, ISF_WIN_WAVES, true, gen != 0);
        if (w > bb) bb = w;
    }
    const int b = (int)bb;
    hipError_t e = hipSuccess;
#define ISF_SET(INV, NVR)                                                                         \
    if (e == hipSuccess)                                                                          \
    e = hipFuncSetAttribute((const void*)k_dp_unary_fast<INV, NVR, false>,                        \
                            hipFuncAttributeMaxDynamicSharedMemorySize, b);                       \
    if (e == hipSuccess)                                                                          \
    e = hipFuncSetAttribute((const void*)k_dp_unary_fast<INV, NVR, true>,                         \
                            hipFuncAttributeMaxDynamicSharedMemorySize, b);                       \
    if (e == hipSuccess)                                                                          \
    e = hipFuncSetAttribute((const void*)k_dp_unary_fast<INV, NVR, false, true>,                  \
                            hipFuncAttributeMaxDynamicSharedMemorySize, b);                       \
    /* (the GEN, LUTF and REPAIR forms of the windowed launch: same LDS layout, bounded by b) */   \
    if (e == hipSuccess)                                                                          \
    e = hipFuncSetAttribute((const void*)k_dp_unary_fast<INV, NVR, false, true, true>,            \
                            hipFuncAttributeMaxDynamicSharedMemorySize, b);                       \
    if (e == hipSuccess)                                                                          \
    e = hipFuncSetAttribute((const void*)k_dp_unary_fast<INV, NVR, false, true, false, true>,     \
                            hipFuncAttributeMaxDynamicSharedMemorySize, b);                       \
    if (e == hipSuccess)                                                                          \
    e = hipFuncSetAttribute((const void*)k_dp_unary_fast<INV, NVR, false, true, false, false, true>, \
                            hipFuncAttributeMaxDynamicSharedMemorySize, b)
    if (nvr == 2) { ISF_SET(true, 2); ISF_SET(false, 2); } else { ISF_SET(true, 4); ISF_SET(false, 4); }
#undef ISF_SET
    return e;
}

/* 1 when this call's FAST columns all run the carry-only (GEN) instantiation: every tile windowed, four waves per
 * workgroup (the row residues of the rebuild), the ring kernel in use.  The prepare launch then stores the carry
 * rows of lutT only (DevParams::lut_carry). */
int isk_unary_uses_carry(const DevParams* P, int ncols) {
    /* opt-in (IS_LUT_CARRY=1): MEASURED on MI355X, 64 frames of 1024x2048x128, unary, bit-exact: the prepare launch
     * 2.61 -> 1.39 ms, but this kernel 4.57 -> 8.09 ms (5.69 with the exact entries stubbed out): 6575 instead of 8816
     * images/s.  See DESIGN.md section 6, "carry rows only". */
    if (!ISF_GEN || P->knob_lut_carry != 1 || ISF_WIN_WAVES != 4 || IS_P1_WIN != 32) return 0;
    if (isk_unary_fast_chunk_rows(P) == 0 || P->knob_ring_kernel == 0 || P->knob_unary_diag != 0) return 0;
    if (!IS_P1_WINDOWED(P->D) || P->win_lo == nullptr) return 0;
    if (!(P->knob_win_tiles >= 0 || ncols >= ISF_WIN_MIN_COLS)) return 0;
    return P->win_tiles >= P->ntiles ? 1 : 0;
}

/* 1 when this call's LUT units run inside the DP launch (LUTF): every tile windowed in ONE launch of 4-wave
 * workgroups, 1, 2 or 4 units per column (a LUT block is four waves). */
int isk_unary_uses_fused_lut(const DevParams* P, int ncols) {
    const int fnb = (P->D + 63) / 64;
    if (P->knob_lut_fused == 0 || P->lut_ready == nullptr || P->lutf_bad == nullptr || ISF_WIN_WAVES != 4 ||
        (fnb != 1 && fnb != 2 && fnb != 4))
        return 0;
    if (P->knob_lut_carry == 1) return 0;
    if (isk_unary_fast_chunk_rows(P) == 0 || P->knob_ring_kernel == 0 || P->knob_unary_diag != 0) return 0;
    if (!IS_P1_WINDOWED(P->D) || P->win_lo == nullptr) return 0;
    if (!(P->knob_win_tiles >= 0 || ncols >= ISF_WIN_MIN_COLS)) return 0;
    if (P->win_tiles < P->ntiles) return 0;
    /* by itself only where it pays (frames/s fused | prepare launch at 1 / 4 / 8 / 12 / 16 / 64 frames per call: 4730 |
     * 6240, 7070 | 7100, 9300 | 9060, 9930 | 9440, 9610 | 9100, 11 030 | 10 160); IS_LUT_FUSED=1 / 2: at any size */
    const bool by_itself = P->knob_lut_fused < 0 || P->knob_lut_fused == 3; /* (3, tests: the default policy with a wrong XCC id published) */
    if (by_itself && (ncols < ISF_LUTF_MIN_COLS || fnb > 2)) return 0; /* (D = 256, four units per column, 32 frames of 1024x4096: 3940 | 4040) */
    return P->knob_lut_fused >= 2 ? 2 : 1;
}

/* FAST columns of the batch; the caller runs k_dp_unary<.., false> for the generic ones. */
hipError_t isk_launch_dp_unary_fast(const DevParams* P, int ncols, const RowRec* recs,
                                    const float* lutT, const float* rcp, const int* vhor,
                                    const int* col_flags, const PruneRec* prune, float* cost_table,
                                    int32_t* index_table, unsigned long long* counters,
                                    const float* joined, const float* cost_T, hipStream_t stream) {
    const int nvr = isk_unary_fast_chunk_rows(P);
    /* the diagonal blocks first, two columns per wave (k_dp_unary_diag); needs an even number of
     * columns per image (a pair never straddles two images); IS_UNARY_DIAG=0: the ring kernel walks
     * them itself */
    const int pre_diag = (P->knob_unary_diag != 0 && (P->C % 2) == 0 && (ncols % 2) == 0) ? 1 : 0;
    if (pre_diag) {
        const dim3 dgrid((unsigned)(ncols / 2) * (unsigned)P->ntiles);
        if (P->invalid >= 0)
            hipLaunchKernelGGL(k_dp_unary_diag<true>, dgrid, dim3(64), 0, stream, *P, ncols, recs, lutT, joined,
                               rcp, vhor, col_flags, cost_table, index_table);
        else
            hipLaunchKernelGGL(k_dp_unary_diag<false>, dgrid, dim3(64), 0, stream, *P, ncols, recs, lutT, joined,
                               rcp, vhor, col_flags, cost_table, index_table);
    }
    const int groups = (ncols + 7) / 8;
    /* The tiles 0 .. wt - 1 (P->win_tiles: those that start below every horizon of the batch) stage an fn window of
     * IS_P1_WIN lutT columns instead of all D and run as ISF_WIN_WAVES-wave workgroups in a launch of their own:
     * 20 instead of 52 KB of LDS per workgroup, seven 4-wave workgroups per CU instead of three 8-wave ones.  The
     * tiles do not depend on each other: the taller (classic) ones go first. */
    int wt = 0;
    if (!pre_diag && IS_P1_WINDOWED(P->D) && P->win_lo != nullptr &&
        (P->knob_win_tiles >= 0 || ncols >= ISF_WIN_MIN_COLS))
        wt = P->win_tiles < P->ntiles ? P->win_tiles : P->ntiles;
    const int nw_win = ISF_WIN_WAVES;
    const size_t lds = isk_unary_fast_lds_bytes(P, nvr);
    /* LUTF: 8 LUT blocks + cpb 8 ntiles DP blocks per super-group of cpb = 4 / fn_blocks column groups */
    unsigned fused_grid = 0;
    if (P->lut_fused) {
        if (wt < P->ntiles) return hipErrorInvalidValue;
        const int fnb = (P->D + 63) / 64, cpb = 4 / fnb;
        const int nSG = (groups + cpb - 1) / cpb;
        fused_grid = (unsigned)nSG * (8u + (unsigned)(cpb * 8 * P->ntiles));
    }
    const bool gen = P->lut_carry != 0; /* (set by the caller only when isk_unary_uses_carry() holds) */
    if (gen && wt < P->ntiles) return hipErrorInvalidValue;
    const size_t lds_win = isf_lds_bytes(P, nvr, nw_win, true, gen);
#define ISF_LAUNCH(INV, NVR)                                                                      \
    do {                                                                                          \
        if (pre_diag)                                                                             \
            hipLaunchKernelGGL((k_dp_unary_fast<INV, NVR, true>), dim3(groups * 8 * P->ntiles), dim3(ISF_THREADS), \
                               lds, stream, *P,                                                    \
                               ncols, recs, lutT, rcp, vhor, col_flags, prune, cost_table, index_table, \
                               counters, joined, cost_T, pre_diag, 0, P->ntiles, nullptr);                  \
        else {                                                                                    \
            if (wt < P->ntiles)                                                                   \
                hipLaunchKernelGGL((k_dp_unary_fast<INV, NVR, false>), dim3(groups * 8 * (P->ntiles - wt)), \
                                   dim3(ISF_THREADS), lds, stream, *P,                             \
                                   ncols, recs, lutT, rcp, vhor, col_flags, prune, cost_table, index_table, \
                                   counters, joined, cost_T, pre_diag, wt, P->ntiles - wt, nullptr);        \
            if (wt > 0 && gen)                                                                    \
                hipLaunchKernelGGL((k_dp_unary_fast<INV, NVR, false, true, true>), dim3(groups * 8 * wt), \
                                   dim3(nw_win * 64), lds_win, stream, *P,                         \
                                   ncols, recs, lutT, rcp, vhor, col_flags, prune, cost_table, index_table, \
                                   counters, joined, cost_T, pre_diag, 0, wt, nullptr);                     \
            else if (wt > 0 && P->lut_fused) {                                                    \
                hipLaunchKernelGGL((k_dp_unary_fast<INV, NVR, false, true, false, true>), dim3(fused_grid), \
                                   dim3(nw_win * 64), lds_win, stream, *P,                         \
                                   ncols, recs, lutT, rcp, vhor, col_flags, prune, cost_table, index_table, \
                                   counters, joined, cost_T, pre_diag, 0, wt, nullptr);            \
                /* the repair launches: they leave at once unless a workgroup above set lutf_bad */  \
                const hipError_t er = isk_launch_lut_repair(P, ncols, joined, cost_T, const_cast<float*>(lutT), stream); \
                if (er != hipSuccess) return er;                                                  \
                hipLaunchKernelGGL((k_dp_unary_fast<INV, NVR, false, true, false, false, true>),   \
                                   dim3(groups * 8 * wt < 1536 ? groups * 8 * wt : 1536),          \
                                   dim3(nw_win * 64), lds_win, stream, *P,                         \
                                   ncols, recs, lutT, rcp, vhor, col_flags, prune, cost_table, index_table, \
                                   nullptr, joined, cost_T, pre_diag, 0, wt, P->lutf_bad);         \
            }                                                                                     \
            else if (wt > 0)                                                                      \
                hipLaunchKernelGGL((k_dp_unary_fast<INV, NVR, false, true>), dim3(groups * 8 * wt), \
                                   dim3(nw_win * 64), lds_win, stream, *P,                         \
                                   ncols, recs, lutT, rcp, vhor, col_flags, prune, cost_table, index_table, \
                                   counters, joined, cost_T, pre_diag, 0, wt, nullptr);                     \
        }                                                                                         \
    } while (0)
    if (P->invalid >= 0) {
        if (nvr == 2) ISF_LAUNCH(true, 2); else ISF_LAUNCH(true, 4);
    } else {
        if (nvr == 2) ISF_LAUNCH(false, 2); else ISF_LAUNCH(false, 4);
    }
#undef ISF_LAUNCH
    return hipGetLastError();
}

} /* extern "C" */
