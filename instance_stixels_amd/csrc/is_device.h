/*
 * is_device.h -- device-side data layout of the gfx950 column-DP core.
 *
 * HBM layout per stixel column (see DESIGN.md "Data layout"):
 *   RowRec recs[rows+1]      one 128-byte record per prefix index v (0..rows): every quantity
 *                            the DP needs at a segment boundary, so that a segment (vB, vT)
 *                            costs one record at vT+1 (held in VGPRs by the lane that owns
 *                            vT) and one record at vB (wave-uniform -> scalar loads -> SGPRs).
 *   float   lutT[rows+1][D]  object data-cost prefix table, transposed w.r.t. the reference's
 *                            d_object_lut[fn][v] (Stixels.cu:159-160) so that the vB-side
 *                            gather of a wave stays inside one 4*D-byte row.
 */
#ifndef IS_DEVICE_H_
#define IS_DEVICE_H_

#include <stdint.h>

#ifndef IS_UNARY_WAVES
#define IS_UNARY_WAVES 8   /* waves per unary-DP workgroup; x4 workgroups/CU (LDS) = waves/SIMD */
#endif
#ifndef IS_UNARY_OCC
#define IS_UNARY_OCC IS_UNARY_WAVES /* waves per SIMD the unary DP is compiled for (VGPR budget 512/OCC) */
#endif
#ifndef IS_CMPX_UPDATE
#define IS_CMPX_UPDATE 1 /* running minima of the unary DP through v_cmpx + moves (take_if_less) */
#endif
#ifndef IS_PRUNE
#define IS_PRUNE 1 /* exact branch-and-bound on vB in FAST columns (descending vB, early exit) */
#endif
#ifndef IS_SKIP_GROUND_ABOVE_HORIZON
#define IS_SKIP_GROUND_ABOVE_HORIZON 1 /* tiles above the horizon: ground candidates cost +inf, skip them */
#endif
#define IS_TILE 64
/* Rows per BOUND BLOCK of the pairwise branch-and-bound (lemmas L6 / L7): the running
 * minima q of the StepRecs restart, the pre-pass of phase 1 evaluates a block top, and phase 2 leaves
 * a separable summary every IS_QB candidate rows.  A bound loses up to one block of accumulated path
 * cost, so smaller blocks close a type sooner but cost more block tops per tile (64: the tile, round
 * 3; 32 measured best in round 4).  Block j >= 1 = the candidate rows IS_QB (j - 1) + 1 .. IS_QB j,
 * block 0 = the first segment vB = 0. */
#ifndef IS_QB_LOG
#define IS_QB_LOG 5
#endif
#define IS_QB (1 << IS_QB_LOG)
#define IS_QPT (IS_TILE / IS_QB) /* bound blocks per tile */
#define IS_PW_MAX_SPLIT 4            /* phase-1 workgroups per (column, tile) at small batches */
#define IS_PW_SPLIT_TARGET_WGS 1024 /* partial-minima slots reserved for the split phase 1 */
#ifndef IS_PW_SPLIT_MAX_COLS
#define IS_PW_SPLIT_MAX_COLS 512     /* up to that many columns: two phase-1 workgroups per (column, tile) */
#endif
#define IS_PAIRWISE_SPLIT_MIN_COLS 1024 /* columns per group before the pairwise DP uses one more stream */
#define IS_PAIRWISE_MAX_GROUPS 3       /* column groups (streams of the context) of the pairwise DP; IS_PW_GROUPS overrides.  The latency-bound phase 2 of one group runs beside the launches of the others.  Round 5, frames/s with 1 / 2 / 3 / 4 groups: batch 64 4116 / 4283 / 4333 / 4227, batch 32 3776 / 3909 / 4009 / 3992, batch 16 2937 / 3120 / 3147 / 3219 (profiles/r05_ab_groups.log).  Beside a PIPELINED RCCL gather of the previous step's output (bench.py --gpus N, parallel.py) the groups cost 5 % instead (round 4: 3680 against 3860): such callers create their context with IS_PW_GROUPS=1, as bench.py does */
#define IS_P2_SPLIT_MAX_COLS 2048     /* up to eight 2048-px frames: phase 2 of the pairwise DP as chain + evaluator wave per column */
#define IS_BACKTRACE_STAGE_MAX_COLS 2048 /* up to eight 2048-px frames: the back-trace chases in LDS */
#define IS_AUX_STREAMS 7               /* auxiliary streams a context owns */
#define IS_N_ON 8           /* non-instance object classes 2..9   (Cityscapes.h:69) */
#define IS_N_OI 8           /* instance object classes     11..18 (Cityscapes.h:75) */

/* Full-resolution exclusive prefix values at row index v (v = 0..rows).
 * F_c[v] = sum_{j<v} x_c[j/8] in wrapping int32: DownsampledSum(c, vB, vT) of
 * Cityscapes.h:28-42 equals F_c[vT+1] - F_c[vB] exactly (ring arithmetic mod 2^32).
 *
 * Two encodings of the same 128 bytes, chosen per column by k_prepare_columns (col_flags):
 *  - FAST (flag 0): everything is fp32 and every float(integer difference) is one or three
 *    2-cycle fp32 subtractions/additions (on gfx950 integer subtraction with an SGPR operand,
 *    conversions, min/max and f64 all cost 4 cycles per wave):
 *      * the 19 class prefixes as fp32 -- exact, because FAST requires every class value >= 0
 *        and every column total < 2^24, so each prefix and difference is an integer in [0, 2^24);
 *      * sum(mx), sum(my) as fp32 -- exact, because FAST requires sum|mx|, sum|my| < 2^23, so
 *        every prefix is below 2^23 and every difference below 2^24 in magnitude;
 *      * sum(mx^2), sum(my^2) split as P = Ph + Pl with Pl = P mod 2^22: the same bound gives
 *        P <= max|mx| * sum|mx| < 2^46, so Ph (a multiple of 2^22 below 2^46), Pl (< 2^22) and
 *        both differences are exact in fp32, and RN((Ah-Bh) + (Al-Bl)) = RN(A-B) = float(A-B).
 *  - generic (flag != 0): int32 / int64 bit patterns, see RowRecWide. */
struct __attribute__((aligned(128))) RowRec {
    float Fg0, Fg1;          /* classes 0 (road), 1 (sidewalk)            */
    float Fon[IS_N_ON];      /* classes 2..9                                */
    float Foi[IS_N_OI];      /* classes 11..18                              */
    float Fsky;              /* class 10                                    */
    int32_t Fnic;            /* squared offset channels, x + y (StixelsKernels.cu:62-70) */
    float G;                 /* ground data-cost prefix (Blelloch association)  */
    float K;                 /* sky data-cost prefix    (Blelloch association)  */
    float S;                 /* disparity prefix        (Blelloch association)  */
    float V;                 /* valid-pixel count prefix (exact)                */
    /* instance-centre prefix sums (StixelsKernels.cu:401-409), FAST encoding */
    float MX, MY;            /* sum(mx), sum(my)                                 */
    float MX2h, MX2l;        /* sum(mx^2) = MX2h + MX2l, MX2l = sum mod 2^22    */
    float MY2h, MY2l;        /* sum(my^2) likewise                               */
    float pad[2];
};
static_assert(sizeof(RowRec) == 128, "RowRec must be one 128-byte line");

struct __attribute__((aligned(128))) RowRecWide { /* generic encoding of the same bytes */
    int32_t Fg0, Fg1;
    int32_t Fon[IS_N_ON];
    int32_t Foi[IS_N_OI];
    int32_t Fsky;
    int32_t Fnic;
    float G, K, S, V;
    int64_t MX, MY, MX2, MY2;
};
static_assert(sizeof(RowRecWide) == 128, "RowRecWide must alias RowRec");

#define IS_FAST_CLASS_LIMIT (1 << 24)      /* column total of every class channel          */
#define IS_FAST_INSTANCE_LIMIT (1 << 23)   /* sum|mx|, sum|my| bound of FAST columns       */
#define IS_FAST_SPLIT_BITS 22              /* low part of the squared-sum split            */
#define IS_FAST_DISP_MIN 0x1p-60f          /* nonzero |d| range of FAST columns: then every */
#define IS_FAST_DISP_MAX 0x1p60f           /* prefix difference is 0 or in [2^-84, 2^75]    */

/* Per-vB transition priors of the pairwise model that do not depend on the DP state
 * (StixelsKernels.cu:88-199); one 32-byte record per vB, read with scalar loads. */
struct __attribute__((aligned(32))) PriorRec {
    float pc;        /* GetPriorCost(vB, rows)                      :40-42   */
    float g_from;    /* GetPriorCostGround(pc)                      :185-187 */
    float s_from_g;  /* GetPriorCostSkyFromGround                   :98-106  */
    float o_from_s;  /* GetPriorCostObjectFromSky if fn > epsilon   :173-183 */
    float og_hi;     /* GetPriorCostObjectFromGround, fn > g+eps    :132-135 */
    float og_lo;     /*                               fn < g-eps    :136-139 */
    float og_mid;    /*                               otherwise     :140-142 */
    float g_prev;    /* max(0, ground_function[vB-1])               :127-130 */
};
static_assert(sizeof(PriorRec) == 32, "PriorRec must be 32 bytes");

/* Per-column constants of the exact branch-and-bound on vB (DESIGN.md "Pruning").  Every E is
 * a non-negative fp32 slack; +inf in E1o switches the pruning of the column off (every lower
 * bound becomes -inf).  Written by k_prepare_columns, read with scalar loads.
 *   E1x = dw * sigma_x, sigma_x >= -(smallest possible data term of a segment): the data terms are
 *         differences of fp32 tree-summed prefixes of per-row costs; sigma covers negative per-row
 *         costs (nu * H) and the summation error (2 * gamma * sum|x|);
 *   E2  >= -(smallest possible instance term ic): the computed sum(x^2) - (sum x)^2 / h can be
 *         negative only through rounding, by at most 8 * 2^-24 * iw * (sum mx^2 + sum my^2). */
struct __attribute__((aligned(32))) PruneRec {
    float E1o, E1g, E1s, E2;
    float pad[4];
};
static_assert(sizeof(PruneRec) == 32, "PruneRec must be 32 bytes");

/* Evaluation counters (is_set_eval_counters): wave-steps of 64 (vB, vT) pairs the pruned walks of
 * FAST columns actually evaluated BELOW the diagonal blocks (those are always evaluated in full).
 * A null pointer in timed runs; bench.py reads them in a separate, untimed pass. */
#define IS_CNT_UNARY_FULL 0 /* k_dp_unary_fast: full steps (three candidates)          */
#define IS_CNT_UNARY_GS 1   /* k_dp_unary_fast: ground- / sky-only steps               */
#define IS_CNT_P1_FULL 2    /* k_pw_phase1: full steps (incl. the first segment vB = 0) */
#define IS_CNT_P1_GS 3      /* k_pw_phase1: ground- / sky-only candidates               */
#define IS_CNT_P1_WINMISS 4 /* k_pw_phase1: steps in which some lane read outside its fn window (IS_P1_WIN) */
#define IS_CNT_UNARY_WINMISS 5 /* k_dp_unary_fast (windowed tiles): steps in which some lane read outside its fn window */
#define IS_CNT_LUTF_SPINS 6       /* k_dp_unary_fast (LUTF): polls of DP workgroups that found their column's LUT units unfinished */
#define IS_CNT_LUTF_UNIT_CYCLES 7 /* k_dp_unary_fast (LUTF): shader clocks the fused LUT units lived, summed */
#define IS_CNT_TILE0 8    /* + 3 * tile + {0 full, 1 window misses, 2 ground / sky-only}: per phase-1 launch, tile < 64 */
#define IS_CNT_N 200

struct DevParams {
    int H, C, D, P2, P2S, CH, K, S;
    int ntiles;        /* ceil(H / IS_TILE) */
    int log2P2;
    float invalid;
    /* sky / ground data terms (StixelsKernels.cu:201-234) */
    float pnex_sky_log, norm_sky, inv_sigma2_sky, puniform_sky, nopnex_sky_log;
    float pnex_gnd_log, puniform, nopnex_gnd_log;
    /* weights */
    float dw, pw, sw, iw;
    /* pairwise model */
    float rows_log, max_dis_log, epsilon, pgrav, pblg, pord, max_disf;
    float log2c, nlog07, nlog03; /* is_logf(2), -is_logf(0.7), -is_logf(0.3) */
    float nlog_pord, nlog_1mpord; /* -is_logf(pord), -is_logf(1 - pord) (NegFastLogDiv, :162, :166) */
    float first_g;  /* GetPriorCostGroundFirst  :196-199 */
    float first_o_below, first_o_above; /* GetPriorCostObjectFirst :189-194 */
    int size_filter;
    int column_step;
    /* branch-and-bound (see PruneRec): object data-term slack from the host's obj_cost_lut
     * (+inf = pruning disabled: negative / non-finite weights or non-finite table), and
     * 2 * gamma_d, the relative error bound of the tree-summed prefixes */
    float sigma_od;
    float gamma2;
    /* host-side launch knobs: the IS_* environment variables, read ONCE in is_ctx_create (never
     * per call); -1 = automatic.  The kernels ignore them. */
    int knob_ring_kernel;     /* IS_NO_RING_KERNEL=1 -> 0: unary FAST columns through k_dp_unary */
    int knob_prepare_overlap; /* IS_PREPARE_OVERLAP: 0 = two prepare launches in order, 1 = on two streams, 2 / unset = one fused launch */
    int knob_p2_lds_floor;    /* IS_P2_LDS: floor on phase 2's LDS allocation (occupancy throttle) */
    int knob_pw_groups;       /* IS_PW_GROUPS: column groups (streams) of the pairwise DP */
    int knob_p2_split;        /* IS_P2_SPLIT: 1 = k_pw_phase2s, 0 = k_pw_phase2 */
    int knob_p2x;             /* IS_P2X=0: large batches walk phase 2 with k_pw_phase2 (one column per wave) */
    int knob_win_tiles;       /* IS_P1_WIN_TILES: number of phase-1 tiles that stage an fn window (-1: those below the horizon) */
    int knob_pw_waves;        /* IS_PW_WAVES: waves per phase-1 workgroup for every tile (-1: 8, windowed tiles IS_P1_WIN_WAVES) */
    int knob_unary_diag;      /* IS_UNARY_DIAG=1: the diagonal blocks of the unary DP in k_dp_unary_diag (two columns per wave) */
    /* fn windows of the DP kernels (IS_P1_WIN): [n_columns][ntiles] first lutT column of the window a
     * (column, tile) stages in LDS; written by the prepare kernel, device memory of the context */
    int* win_lo;
    int win_tiles; /* the tiles 0 .. win_tiles - 1 of this call stage a window (set per call: unary every tile,
                    * pairwise phase 1 the tiles that start below every horizon of the batch) */
    /* LUT units INSIDE the unary DP launch (is_k_unary_fast.hip, LUTF): device counters [columns] of finished
     * (column, 64 fn) units, zeroed by k_prepare_columns; lut_fused is set per call */
    int* lut_ready;
    /* ... and one word per context that a DP workgroup sets when it cannot trust what it waited for: its column's units
     * ran on another XCD than itself (the hand-over goes through the XCD's L2: the units publish their XCC id with
     * their count) or did not finish within the bound of the poll.  The launches behind the fused one -- the ordinary
     * LUT kernel and the ordinary DP launch, which leave at once while the word is 0 -- then do the call again. */
    int* lutf_bad;
    /* ... and the number of calls of this context whose repair launches have run: a word in pinned, mapped host memory
     * that block 0 of the repair DP launch counts up.  The host reads it (a plain load, nothing waits) when it plans a
     * call: after the first repair the fused launch is off for the rest of the context's life unless IS_LUT_FUSED asks
     * for it by value.  is_lut_fused_repairs() returns it. */
    int* lutf_repairs;
    int knob_lut_fused; /* IS_LUT_FUSED: -1 / 1 = the LUT units run inside the unary DP launch where they can, 0 = never,
                         * 2 = (tests) fused with a WRONG XCC id published: every workgroup distrusts, the repair launches run,
                         * 3 = (tests) the default policy (-1) with the wrong id of 2: the first large call is repaired, and the
                         * context then keeps the table in the prepare launch (DevParams::lutf_repairs) */
    int lut_fused;      /* set per call: 0, 1, or 2 (the test mode) */
    int knob_lut_carry; /* IS_LUT_CARRY=1: carry rows only wherever the DP can rebuild the rest (unary calls whose every
                         * tile runs the windowed ring kernel); default: lutT is materialised (measured faster) */
    int lut_carry;      /* set per call: k_object_lut stores only the rows 32 k of lutT (the carries of its 32-row
                         * blocks, 1/32 of the table); the DP rebuilds the rows it reads (is_k_unary_fast.hip, GEN) */
};

/* fn windows (k_dp_unary_fast, k_pw_phase1).  A (column, tile) workgroup keeps lutT[vT + 1][*] of its 64 rows in
 * LDS: D + 1 floats per row, 33 KB at D = 128 -- which is what limited a CU to three workgroups.  A lane only
 * reads lutT[vT + 1][floor(mean of its segment)], and the means of the segments a tile evaluates lie close
 * together (a road ramp moves by ~12 disparities over 64 rows, an object by ~1), so when D is larger than
 * IS_P1_WIN (and a multiple of 4) the workgroup stages only the IS_P1_WIN columns from win_lo[column][tile] on:
 * 8.4 KB, small workgroups (4 waves), six or seven per CU.  k_prepare picks the start: the window that begins
 * at the tile's smallest disparity (the segments of a lane start BELOW its row, nearer to the camera: larger
 * or equal disparities on a road scene) or the one that ends at its largest, whichever holds more of the
 * tile's rows.  The vB side uses the same window (unary: ring slots of 32 columns + the record; phase 1: 64
 * columns in one register per lane).  A lane whose floor(mean) falls outside reads global memory: exactness
 * never rests on the window. */
#ifndef IS_P1_WIN
#define IS_P1_WIN 32
#endif
#define IS_P1_WINDOWED(D) (IS_P1_WIN > 0 && (D) > IS_P1_WIN && ((D) & 3) == 0)
/* Round 5: the window is TWO halves of IS_P1_WIN / 2 columns, each anywhere (a multiple of 4): win_lo[column][tile]
 * = first column of half A | first column of half B << 16.  A tile of one disparity cluster gets B = A + 16 -- the
 * contiguous window of round 4 --, a tile above the horizon that holds sky (d ~ 0) AND an object one half per
 * cluster (k_prepare picks whichever of the three forms holds most of the tile's rows).  Staged column c (0 ..
 * IS_P1_WIN - 1) is lutT column IS_WIN_COL(w, c); IS_WIN_FIND(w, fn) is the staged column of fn, or negative. */
#ifndef IS_WIN_SPLIT
#define IS_WIN_SPLIT 1 /* 0: contiguous fn windows only (round 4) */
#endif
#define IS_WIN_HALF (IS_P1_WIN / 2)
#define IS_WIN_A(w) ((int)((unsigned)(w) & 0xFFFFu))
#define IS_WIN_B(w) ((int)((unsigned)(w) >> 16))
#define IS_WIN_PACK(a, b) ((int)((unsigned)(a) | ((unsigned)(b) << 16)))
#define IS_WIN_COL(w, c) ((c) < IS_WIN_HALF ? IS_WIN_A(w) + (c) : IS_WIN_B(w) + (c) - IS_WIN_HALF)
#define IS_WIN_FIND(w, fn)                                                                     \
    ((unsigned)((fn) - IS_WIN_A(w)) < (unsigned)IS_WIN_HALF                                     \
         ? (fn) - IS_WIN_A(w)                                                                   \
         : ((unsigned)((fn) - IS_WIN_B(w)) < (unsigned)IS_WIN_HALF ? (fn) - IS_WIN_B(w) + IS_WIN_HALF : -1))

#endif /* IS_DEVICE_H_ */
