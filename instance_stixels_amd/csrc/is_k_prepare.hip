/* is_k_prepare.hip -- per-column boundary records, object data-cost prefix table, pairwise
 * prior tables.  See is_kernels.h. */
#include "is_kernels.h"

/* ====================================================================================== */
/* A4-A6  per-column preparation                                                           */
/* ====================================================================================== */
#define PREP_THREADS 256
#ifndef PREP_STORE_LATE
#define PREP_STORE_LATE 1
#endif
#ifndef PREP_CLASS_BY_WAVE
#define PREP_CLASS_BY_WAVE 1 /* measured: column blocks 1.04 -> 0.95 ms, fused launch 2.56 -> 2.48 ms per batch of 64 */
#endif
#ifndef PREP_EPI_ROWS
#define PREP_EPI_ROWS 256 /* rows per block of the record epilogue (0 = piece by piece) */
#endif

/* LDS stride of a segmentation channel: H/8 + 1 prefix entries, rounded up to 16 bytes */
__host__ __device__ static inline int prep_seg_stride(int H) { return (((H >> 3) + 1) + 3) & ~3; }
/* leaves of the fp32 scan trees: H when H is a power of two (then P2 = 2H), else P2 */
__host__ __device__ static inline int prep_scan_leaves(int H, int P2) {
    return ((H & (H - 1)) == 0 && P2 == 2 * H && H >= 16) ? H : P2;
}

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

/* (sum, min) over the block; s_red holds 8 floats.  Only used for slack bounds (PruneRec), which
 * carry their own safety factor, so the summation order does not matter. */
__device__ __forceinline__ float2 block_sum_min(float s, float m, float* s_red) {
#pragma unroll
    for (int j = 32; j >= 1; j >>= 1) {
        s += __shfl_xor(s, j, 64);
        m = __builtin_fminf(m, __shfl_xor(m, j, 64));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) { s_red[wave] = s; s_red[4 + wave] = m; }
    __syncthreads();
    float2 r;
    r.x = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    r.y = __builtin_fminf(__builtin_fminf(s_red[4], s_red[5]), __builtin_fminf(s_red[6], s_red[7]));
    return r;
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

#ifndef PREP_REC_NT
#define PREP_REC_NT 0
#endif
/* one 16-byte piece of a record */
__device__ __forceinline__ void prep_store16(int4* dst, int4 x) {
#if PREP_REC_NT
    typedef int v4i __attribute__((ext_vector_type(4)));
    v4i y = {x.x, x.y, x.z, x.w};
    __builtin_nontemporal_store(y, reinterpret_cast<v4i*>(dst));
#else
    *dst = x;
#endif
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
    prep_store16(d + 6, a);
    prep_store16(d + 7, b);
}

__device__ __forceinline__ void prepare_columns_body(
    const DevParams& P, const int colg, char* smem, const float* __restrict__ joined,
    const int32_t* __restrict__ seg, const float* __restrict__ ground /*[img][3][H]*/,
    const int* __restrict__ vhor_arr, RowRec* __restrict__ recs, int* __restrict__ col_flags,
    float* __restrict__ sv_arr, PruneRec* __restrict__ prune, int* __restrict__ n_generic) {
    const int H = P.H, P2 = P.P2, P2S = P.P2S, CH = P.CH, K = P.K;
    if (P.lut_ready != nullptr && threadIdx.x == 0) {
        P.lut_ready[colg] = 0; /* (the fused LUT units of the DP launch count up) */
        if (colg == 0 && P.lutf_bad != nullptr) *P.lutf_bad = 0;
    }
    /* LDS stride of a segmentation channel: only its first H/8 + 1 entries matter (an exclusive
     * prefix at index <= H/8 never sees the zero padding up to P2S), and 21 channels of P2S = 256
     * entries were 21.5 of the kernel's 46 KB of LDS: with 132 the CU holds four workgroups, not three */
    const int SS = prep_seg_stride(H);
    /* The scans run over P2 >= H + 1 zero-padded elements (StixelsKernels.h:73-103).  When H is a
     * power of two (P2 = 2H) everything a prefix at index <= H needs lies in the LEFT half of that
     * tree: indices < H walk the same nodes as in a tree over H leaves, and index H is the left
     * child of the root = the root of the H-leaf tree.  The tree is then built over NP = H leaves:
     * identical sums, half the LDS (12 instead of 24 KB at H = 1024). */
    const int NP = prep_scan_leaves(H, P2);
    const int LNP = (NP == P2) ? P.log2P2 : P.log2P2 - 1;
    float* s_d = (float*)smem;                          /* [NP]   disparity column        */
    float* s_pyr = s_d + NP;                            /* [2*NP] scan tree                */
    int32_t* s_seg = (int32_t*)(s_pyr + 2 * NP);        /* [CH][SS]                        */
    auto prefix_at = [&](int v) -> float { /* exclusive prefix at v <= H */
        return (v == NP) ? s_pyr[2 * NP - 2] : blelloch_prefix(s_pyr, NP, LNP, v);
    };
    int64_t* s_wave = (int64_t*)(s_seg + CH * SS);      /* [4]                             */
    float* s_red = (float*)(s_wave + 4);                /* [8] block reductions            */
    float* s_tot = s_red + 8;                           /* [2] sum mx^2 + my^2 of the column */

    const int img = colg / P.C, col = colg % P.C;
    const int vhor = vhor_arr[img];
    const float* gfun = ground + (size_t)img * 3 * H;
    const float* gnorm = gfun + H;
    const float* gis2 = gnorm + H;
    const float* dcol = joined + (size_t)colg * H;
    const int32_t* scol = seg + (size_t)colg * CH * P2S;
    RowRec* rcol = recs + (size_t)colg * (H + 1);
    const int tid = threadIdx.x;

    /* column inputs into LDS with 16-byte loads (H is a multiple of 8; the segmentation column is
     * 16-byte aligned when P2S is a multiple of 4) */
    {
        const float4* d4 = reinterpret_cast<const float4*>(dcol);
        float4* sd4 = reinterpret_cast<float4*>(s_d);
        for (int i = tid; i < (H >> 2); i += PREP_THREADS) sd4[i] = d4[i];
        for (int i = H + tid; i < NP; i += PREP_THREADS) s_d[i] = 0.0f;
        if ((P2S & 3) == 0 && SS <= P2S) {
            const int4* g4 = reinterpret_cast<const int4*>(scol);
            int4* s4 = reinterpret_cast<int4*>(s_seg);
            const int sq = SS >> 2, gq = P2S >> 2;
#pragma unroll 4
            for (int i = tid; i < CH * sq; i += PREP_THREADS) {
                const int c = i / sq, q = i - c * sq;
                s4[i] = g4[c * gq + q];
            }
        } else {
            for (int i = tid; i < CH * SS; i += PREP_THREADS) {
                const int c = i / SS, k = i - c * SS;
                s_seg[i] = k < P2S ? scol[c * P2S + k] : 0;
            }
        }
    }
    __syncthreads();

    /* ---- fn windows of the pairwise phase 1 (is_device.h, IS_P1_WIN): per 64-row tile the smallest valid
     * disparity, rounded down to a multiple of 4 columns (16-byte loads), one below for the rounding of
     * the prefix sums a mean is computed from */
    if (IS_P1_WINDOWED(P.D) && P.win_lo != nullptr) {
        const int lane = tid & 63, wv = tid >> 6;
        for (int t = wv; t < P.ntiles; t += PREP_THREADS / 64) {
            const int r = t * 64 + lane;
            float d = IS_INF, dx = -IS_INF;
            bool ok = false;
            if (r < H) {
                const float x = s_d[r];
                if (!(P.invalid >= 0 && x == P.invalid) && x == x) { d = dx = x; ok = true; }
            }
            const float mine = d;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                d = __builtin_fminf(d, __shfl_xor(d, m, 64));
                dx = __builtin_fmaxf(dx, __shfl_xor(dx, m, 64));
            }
            /* three candidates: the contiguous window that starts at the tile's smallest disparity (the segments of
             * a lane start below its row -- nearer, larger disparities), the contiguous one that ends at its largest,
             * and the SPLIT one: half A from the smallest disparity on, half B up to the largest -- a tile above the
             * horizon can hold sky (d ~ 0) AND an object.  Whichever holds most of the tile's rows (ties: contiguous). */
            constexpr int HW = IS_WIN_HALF;
            int lo_a = 0, lo_b = 0, hi_b = 0;
            if (d < IS_INF) {
                lo_a = (int)__builtin_fminf(__builtin_fmaxf(d, 1.0f), (float)P.D) - 1;
                lo_b = (int)__builtin_fminf(__builtin_fmaxf(dx, 0.0f), (float)(P.D - 1)) + 2 - IS_P1_WIN;
                hi_b = (int)__builtin_fminf(__builtin_fmaxf(dx, 0.0f), (float)(P.D - 1)) + 2 - HW;
            }
            lo_a = min(max(lo_a, 0) & ~3, P.D - IS_P1_WIN);
            lo_b = min(max(lo_b, 0) & ~3, P.D - IS_P1_WIN);
            hi_b = min(max(hi_b, 0) & ~3, P.D - HW);
            const int w_a = IS_WIN_PACK(lo_a, lo_a + HW), w_b = IS_WIN_PACK(lo_b, lo_b + HW);
            const int w_s = IS_WIN_PACK(lo_a, max(hi_b, lo_a + HW)); /* (never overlapping halves) */
            const int fl = (int)__builtin_fminf(__builtin_fmaxf(mine, 0.0f), (float)(P.D - 1));
            const int n_a = __builtin_popcountll(__builtin_amdgcn_ballot_w64(ok && IS_WIN_FIND(w_a, fl) >= 0));
            const int n_b = __builtin_popcountll(__builtin_amdgcn_ballot_w64(ok && IS_WIN_FIND(w_b, fl) >= 0));
            const int n_s = __builtin_popcountll(__builtin_amdgcn_ballot_w64(ok && IS_WIN_FIND(w_s, fl) >= 0));
            int w_best = n_b > n_a ? w_b : w_a;
            if (IS_WIN_SPLIT && n_s > max(n_a, n_b)) w_best = w_s;
            if (lane == 0) P.win_lo[(size_t)colg * P.ntiles + t] = w_best;
        }
    }

    /* ---- instance-centre values per row from the RAW offsets (StixelsKernels.cu:401-409);
     * thread t owns rows [t*R, t*R+R).  mx = 8*col + 3.5 + offx + 0.5 is an exact integer;
     * my = trunc(row - offy + 0.5): n for n >= 0, n + 1 for n < 0 (truncation toward zero). */
    const int R = (H + PREP_THREADS - 1) / PREP_THREADS;
    const int r_lo = tid * R;
    const int32_t* offy = s_seg + K * SS;
    const int32_t* offx = s_seg + (K + 1) * SS;
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
    { /* class channels of a FAST column: values >= 0, full-resolution total < 2^24.  One wave per
       * channel (round robin), a lane sums every 64th entry, the wave adds the lane sums (the
       * nineteen channels used to be walked by nineteen lanes of wave 0: 132 serial iterations
       * with an integer modulo each, a tenth of the workgroup's life) */
        const int lane = tid & 63, wv = tid >> 6;
        for (int c = wv; c < K; c += PREP_THREADS / 64) {
            const int32_t* ch = s_seg + c * SS;
            uint64_t total = 0;
            int negative = 0;
            for (int k = lane; k < SS; k += 64) {
                const int32_t x = ch[k];
                negative |= (x < 0);
                total += (uint64_t)(uint32_t)x;
            }
#pragma unroll
            for (int j = 32; j >= 1; j >>= 1) {
                total += (uint64_t)__shfl_xor((unsigned long long)total, j, 64);
                negative |= __shfl_xor(negative, j, 64);
            }
            slow |= negative | (total * IS_DOWNSAMPLE_FACTOR >= (uint64_t)IS_FAST_CLASS_LIMIT);
        }
    }
    slow = __syncthreads_or(slow);
    if (tid == 0) {
        col_flags[colg] = slow;
        if (slow) atomicAdd(n_generic, 1); /* (zeroed by the caller before the launch) */
    }
    int64_t base_mx = block_excl_scan_i64(sum_mx, s_wave);
    int64_t base_my = block_excl_scan_i64(sum_my, s_wave);
    int64_t base_mx2 = block_excl_scan_i64(sum_mx2, s_wave);
    int64_t base_my2 = block_excl_scan_i64(sum_my2, s_wave);
    /* the owner of rows [r_lo, r_lo+R) writes the exclusive prefix at those indices; the owner
     * of row H-1 also writes index H (the total).  PREP_STORE_LATE: every piece of a 128-byte record
     * (instance prefixes, class chunks, the four fp32 prefixes) is stored at the END of the kernel,
     * back to back, so that the pieces of a line meet in the L2 instead of reaching the memory as
     * eight partial writes spread over the workgroup's life. */
    /* (the raw offsets from the input tensor: the LDS copies are squared in place further down) */
    auto offx_raw = [&](int i) -> int32_t { return scol[(K + 1) * P2S + i]; };
    auto offy_raw = [&](int i) -> int32_t { return scol[K * P2S + i]; };
    /* (dst_row(r): where the record of row r goes) */
    auto instance_rows = [&](bool store, auto dst_row) {
        int64_t bx = base_mx, by = base_my, bx2 = base_mx2, by2 = base_my2;
        for (int r = r_lo; r < r_lo + R && r < H; r++) {
#ifndef PREP_ABL_NOSTORE_INST
            if (store) store_instance_prefix(dst_row(r), slow, bx, by, bx2, by2);
#endif
            const double fx = ((double)(P.column_step * col) + 0.5 * ((double)P.column_step - 1.0)) +
                              (double)offx_raw(r >> 3) + 0.5;
            const int64_t mx = (int64_t)fx;
            const int32_t n32 = (int32_t)((uint32_t)r - (uint32_t)offy_raw(r >> 3));
            const int64_t my = (int64_t)((double)n32 + 0.5);
            bx += mx;
            by += my;
            bx2 = (int64_t)((uint64_t)bx2 + (uint64_t)mx * (uint64_t)mx);
            by2 = (int64_t)((uint64_t)by2 + (uint64_t)my * (uint64_t)my);
        }
        if (r_lo <= H - 1 && H - 1 < r_lo + R) {
            if (store) store_instance_prefix(dst_row(H), slow, bx, by, bx2, by2);
            s_tot[0] = (float)((double)bx2 + (double)by2); /* column totals (PruneRec.E2) */
        }
    };
#ifndef PREP_REC_NT
#define PREP_REC_NT 0
#endif
#ifdef PREP_ABL_RECWRAP /* ablation: the same record stores into 8 rows per column (absorbed by the L2) */
#define PREP_RROW(v) ((v) & 7)
#else
#define PREP_RROW(v) (v)
#endif
    auto rec_row = [&](int r) -> RowRec* { return rcol + PREP_RROW(r); };
    if (PREP_STORE_LATE) {
        /* (the records are stored in the epilogue; here only the column totals are needed, and the owner of row
         * H - 1 has them: its exclusive prefix + its own rows -- the integers instance_rows would end with) */
        if (r_lo <= H - 1 && H - 1 < r_lo + R)
            s_tot[0] = (float)((double)(int64_t)((uint64_t)base_mx2 + (uint64_t)sum_mx2) +
                               (double)(int64_t)((uint64_t)base_my2 + (uint64_t)sum_my2));
    } else {
        instance_rows(true, rec_row);
    }
    __syncthreads();

    /* ---- square the offset channels in place (StixelsKernels.cu:411-416), then exclusive
     * prefix of every channel at 1/8 resolution (:462-469); integer, so any order is exact */
    for (int i = tid; i < 2 * SS; i += PREP_THREADS) {
        const uint32_t x = (uint32_t)s_seg[K * SS + i];
        s_seg[K * SS + i] = (int32_t)(x * x);
    }
    __syncthreads();
    /* branch-and-bound precondition: the non-instance offset term is >= 0 and monotone when no
     * squared entry is negative as int32 (the reference squares in wrapping int32) and the
     * full-resolution total stays below 2^31 */
    int nic_bad = 0;
    {
        unsigned long long sq_sum = 0;
        for (int i = tid; i < 2 * SS; i += PREP_THREADS) {
            const int32_t v = s_seg[K * SS + i];
            nic_bad |= v < 0;
            sq_sum += (unsigned long long)(uint32_t)v;
        }
        unsigned long long* s_abs = (unsigned long long*)s_wave;
        if (tid == 0) s_abs[0] = 0ull;
        __syncthreads();
        atomicAdd(&s_abs[0], sq_sum);
        __syncthreads();
        nic_bad |= (s_abs[0] * IS_DOWNSAMPLE_FACTOR) >= (1ull << 31);
        __syncthreads();
    }
    nic_bad = __syncthreads_or(nic_bad);
    /* one wave per channel (round robin): lane l owns `per` consecutive entries of the SS */
    {
        const int lane = tid & 63, wv = tid >> 6, per = (SS + 63) >> 6;
        for (int c = wv; c < CH; c += PREP_THREADS / 64) {
            int32_t* ch = s_seg + c * SS;
            uint32_t local = 0;
            for (int k = 0; k < per; k++) {
                const int idx = lane * per + k;
                if (idx < SS) local += (uint32_t)ch[idx];
            }
            uint32_t inc = local; /* inclusive wave scan of the lane totals */
#pragma unroll
            for (int j = 1; j < 64; j <<= 1) {
                const uint32_t n = (uint32_t)__shfl_up((int)inc, j, 64);
                if (lane >= j) inc += n;
            }
            uint32_t run = inc - local;
            for (int k = 0; k < per; k++) {
                const int idx = lane * per + k;
                if (idx < SS) {
                    const uint32_t x = (uint32_t)ch[idx];
                    ch[idx] = (int32_t)run;
                    run += x;
                }
            }
        }
    }
    __syncthreads();
    /* one (1/8-resolution block kb, chunk q) item of the class prefixes: emit(v, q, chunk) for its rows */
    auto class_item = [&](int kb, int q, auto emit) {
        const int kn = min(kb + 1, SS - 1); /* (the last block has one row: its increment is never used) */
        uint32_t cur[4], dif[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int dw = q * 4 + j; /* dword of RowRec: Fg0 Fg1 Fon[8] Foi[8] Fsky Fnic */
            if (dw == 19) {
                const int32_t* px = s_seg + (K + 1) * SS;
                const int32_t* py = s_seg + K * SS;
                const uint32_t ax = (uint32_t)px[kb], ay = (uint32_t)py[kb];
                cur[j] = ax * 8u + ay * 8u;
                dif[j] = ((uint32_t)px[kn] - ax) + ((uint32_t)py[kn] - ay);
            } else {
                const int chn = dw < 10 ? dw : (dw < 18 ? dw + 1 : 10);
                const int32_t* ps = s_seg + chn * SS;
                const uint32_t a = (uint32_t)ps[kb];
                cur[j] = a * 8u;
                dif[j] = (uint32_t)ps[kn] - a;
            }
        }
        const bool is_count = (q == 4); /* dword 19 (Fnic) stays an integer in both encodings */
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const int v = kb * 8 + m;
            if (v <= H) {
                int32_t x[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int32_t f = (int32_t)cur[j];
                    x[j] = (slow || (is_count && j == 3)) ? f : __float_as_int((float)f);
                    cur[j] += dif[j];
                }
                emit(v, q, make_int4(x[0], x[1], x[2], x[3]));
            }
        }
    };
    auto class_chunks = [&]() {
    /* dwords 0..19 of every record (class prefixes + squared-offset prefix) as five 16-byte
     * chunks.  A thread owns one chunk of the EIGHT rows of a 1/8-resolution block: the full-resolution
     * prefix at row 8 kb + m is ps[kb] * 8 + (ps[kb + 1] - ps[kb]) * m (full_prefix, wrapping
     * uint32 arithmetic), so the two table entries are read once per block and the rows follow by
     * repeated addition -- an eighth of the LDS reads and a fifth of the instructions of one
     * (row, chunk) item per thread. */
    {
        const int NB = (H >> 3) + 1; /* blocks; the last one holds row H only */
        for (int it = tid; it < NB * 5; it += PREP_THREADS) {
            const int kb = it / 5, q = it - kb * 5;
            class_item(kb, q, [&](int v, int qq, int4 x) {
#ifndef PREP_ABL_NOSTORE_CLASS
                prep_store16(reinterpret_cast<int4*>(rcol + PREP_RROW(v)) + qq, x);
#endif
            });
        }
    }
    };
    if (!PREP_STORE_LATE) class_chunks();

    /* ---- fp32 prefixes with the reference's block-scan association (:452-461).  A thread keeps
     * the four prefixes of its rows (v = tid + k * PREP_THREADS) and writes dwords 20..23 of the
     * record {G, K, S, V} as ONE 16-byte store per row at the end. */
    constexpr int MAXR = 9; /* rows per thread held in registers: H + 1 <= 9 * 256 */
    const bool regs = (H + 1) <= MAXR * PREP_THREADS;
    const bool epi = PREP_EPI_ROWS > 0 && PREP_STORE_LATE && regs && R > 0 && (PREP_EPI_ROWS % (R > 0 ? R : 1)) == 0;
    float pS[MAXR], pV[MAXR], pG[MAXR], pK[MAXR];
#pragma unroll
    for (int k = 0; k < MAXR; k++) pS[k] = pV[k] = pG[k] = pK[k] = 0.0f;
    float* svcol = sv_arr + (size_t)colg * 2 * (H + 1); /* compact copies for the pairwise phase 2 */
    /* S: disparity (valid-masked when invalid >= 0, :382-389) */
    for (int i = tid; i < NP; i += PREP_THREADS) {
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
    blelloch_build(s_pyr, NP, LNP);
    if (regs) {
#pragma unroll
        for (int k = 0; k < MAXR; k++) {
            const int v = tid + k * PREP_THREADS;
            if (v <= H) { pS[k] = prefix_at(v); svcol[v] = pS[k]; }
        }
    } else {
        for (int v = tid; v <= H; v += PREP_THREADS) {
            const float x = prefix_at(v);
            rcol[v].S = x;
            svcol[v] = x;
        }
    }
    __syncthreads();
    /* V: valid count (all zero without an invalid-disparity value: no scan needed) */
    if (P.invalid >= 0) {
        for (int i = tid; i < NP; i += PREP_THREADS)
            s_pyr[i] = (i < H) ? (float)(s_d[i] != P.invalid) : 0.0f;
        __syncthreads();
        blelloch_build(s_pyr, NP, LNP);
        if (regs) {
#pragma unroll
            for (int k = 0; k < MAXR; k++) {
                const int v = tid + k * PREP_THREADS;
                if (v <= H) { pV[k] = prefix_at(v); svcol[H + 1 + v] = pV[k]; }
            }
        } else {
            for (int v = tid; v <= H; v += PREP_THREADS) {
                const float x = prefix_at(v);
                rcol[v].V = x;
                svcol[H + 1 + v] = x;
            }
        }
        __syncthreads();
    } else {
        for (int v = tid; v <= H; v += PREP_THREADS) {
            if (!regs) rcol[v].V = 0.0f;
            svcol[H + 1 + v] = 0.0f;
        }
    }
    /* G: ground data cost, +inf at / above the horizon (:435-446) */
    float g_abs = 0.0f, g_min = 0.0f; /* over the finite rows: slack of the ground data term */
    for (int i = tid; i < NP; i += PREP_THREADS) {
        float x = 0.0f;
        if (i < H) {
            x = (i >= vhor) ? IS_INF : data_cost_ground(gfun[i], s_d[i], gnorm[i], gis2[i], P);
            if (i < vhor) { g_abs += __builtin_fabsf(x); g_min = __builtin_fminf(g_min, x); }
        }
        s_pyr[i] = x;
    }
    const float2 g_red = block_sum_min(g_abs, g_min, s_red);
    __syncthreads();
    blelloch_build(s_pyr, NP, LNP);
    if (regs) {
#pragma unroll
        for (int k = 0; k < MAXR; k++) {
            const int v = tid + k * PREP_THREADS;
            if (v <= H) pG[k] = prefix_at(v);
        }
    } else {
        for (int v = tid; v <= H; v += PREP_THREADS) rcol[v].G = prefix_at(v);
    }
    __syncthreads();
    /* K: sky data cost, 0 below the horizon (:424-433) */
    float k_abs = 0.0f, k_min = 0.0f;
    for (int i = tid; i < NP; i += PREP_THREADS) {
        float x = 0.0f;
        if (i < H) x = (i < vhor) ? 0.0f : data_cost_sky(s_d[i], P);
        k_abs += __builtin_fabsf(x);
        k_min = __builtin_fminf(k_min, x);
        s_pyr[i] = x;
    }
    const float2 k_red = block_sum_min(k_abs, k_min, s_red);
    if (tid == 0) {
        /* PruneRec: see is_device.h.  A prefix difference P[a] - P[b] of per-row values x >= -nu
         * with sum|x| = T is >= -(nu * H + gamma2 * T); NaN anywhere makes the slack NaN, and a NaN
         * slack makes every bound NaN, which never passes the `>` test: no pruning. */
        const float safe = 1.0f + 0x1p-10f;
        const float hf = (float)H;
        const float sig_g = ((0.0f - g_red.y) * hf + P.gamma2 * g_red.x) * safe;
        const float sig_k = ((0.0f - k_red.y) * hf + P.gamma2 * k_red.x) * safe;
        PruneRec pr;
        pr.E1o = P.dw * P.sigma_od;
        pr.E1g = P.dw * sig_g;
        pr.E1s = P.dw * sig_k;
        pr.E2 = P.iw * (0x1p-21f * safe) * s_tot[0]; /* 8 * 2^-24 * (sum mx^2 + sum my^2) */
        /* 0 * inf above would be NaN = off as well; make the "off" states explicit */
        if (slow || nic_bad || !(P.sigma_od < IS_INF) || !(pr.E1g < IS_INF) || !(pr.E1s < IS_INF) ||
            !(pr.E2 < IS_INF))
            pr.E1o = pr.E1g = pr.E1s = IS_INF; /* every type: the bounds are tested one by one */
        pr.pad[0] = pr.pad[1] = pr.pad[2] = pr.pad[3] = 0.0f;
        prune[colg] = pr;
    }
    __syncthreads();
    blelloch_build(s_pyr, NP, LNP);
    if (regs) {
#pragma unroll
        for (int k = 0; k < MAXR; k++) {
            const int v = tid + k * PREP_THREADS;
            if (v <= H) {
                pK[k] = prefix_at(v);
#ifndef PREP_ABL_NOSTORE_F4
                if (!epi) prep_store16(reinterpret_cast<int4*>(rcol + PREP_RROW(v)) + 5, make_int4(__float_as_int(pG[k]), __float_as_int(pK[k]), __float_as_int(pS[k]), __float_as_int(pV[k])));
#endif
            }
        }
    } else {
        for (int v = tid; v <= H; v += PREP_THREADS) rcol[v].K = prefix_at(v);
    }
    /* (round 4, measured and removed: the records staged 64 rows at a time in LDS and stored as whole
     * 128-byte lines, eight lanes per record -- 1.53 instead of 1.21 ms for the column blocks of a batch of
     * 64: the 34 extra barriers cost more than the partial lines do; the kernel is bound by its LDS /
     * VALU work, not by its 2.15 GB of stores) */
    if (epi) {
        /* PREP_EPI_ROWS: the epilogue walks the column in blocks of rows and every producer stores its
         * pieces of the block's records before anyone goes on to the next block -- no barrier, the waves
         * only have to stay roughly together: the partial lines a workgroup has in flight are
         * PREP_EPI_ROWS x 128 bytes instead of the whole column's 131 KB (x 128 workgroups per XCD: 16.8 MB
         * against 4 MB of L2 -- lines left the L2 before their last piece arrived). */
        constexpr int ER = PREP_EPI_ROWS > 0 ? PREP_EPI_ROWS : PREP_THREADS;
        constexpr int M = ER >= PREP_THREADS ? ER / PREP_THREADS : 1; /* register slots per block */
        constexpr int SUBS = ER >= PREP_THREADS ? 1 : PREP_THREADS / ER; /* blocks per register slot */
        static_assert(ER % 8 == 0 && (ER >= PREP_THREADS ? ER % PREP_THREADS == 0 : PREP_THREADS % ER == 0),
                      "PREP_EPI_ROWS: a multiple of 8 that divides PREP_THREADS or is a multiple of it");
#pragma unroll
        for (int kk = 0; kk < (MAXR + M - 1) / M; kk++) {
            for (int sub = 0; sub < SUBS; sub++) {
                const int row0 = kk * M * PREP_THREADS + sub * ER;
                if (row0 <= H) {
#pragma unroll
                    for (int j = 0; j < M; j++) { /* {G, K, S, V}: row k * PREP_THREADS + tid is this thread's slot k */
                        const int k = kk * M + j;
                        if (k < MAXR) {
                            const int v = k * PREP_THREADS + tid;
                            if ((SUBS == 1 || tid / ER == sub) && v <= H)
                                prep_store16(reinterpret_cast<int4*>(rcol + PREP_RROW(v)) + 5,
                                             make_int4(__float_as_int(pG[k]), __float_as_int(pK[k]),
                                                       __float_as_int(pS[k]), __float_as_int(pV[k])));
                        }
                    }
#ifdef PREP_ABL_WAVEROWS /* timing-only: every wave stores ALL pieces of its own 64 rows (instance pieces: dummies) */
                    {
                        const int v = kk * M * PREP_THREADS + tid;
                        if (v <= H) {
                            prep_store16(reinterpret_cast<int4*>(rcol + v) + 6, make_int4(v, tid, 0, 0));
                            prep_store16(reinterpret_cast<int4*>(rcol + v) + 7, make_int4(v, tid, 1, 0));
                        }
                        const int ln = tid & 63, wv = tid >> 6;
                        if (ln < 40) {
                            const int kbl = ln / 5, q = ln - kbl * 5, kb = ((row0 + 64 * wv) >> 3) + kbl;
                            if (kb * 8 <= H)
                                class_item(kb, q, [&](int vv, int qq, int4 x) {
                                    prep_store16(reinterpret_cast<int4*>(rcol + PREP_RROW(vv)) + qq, x);
                                });
                        }
                    }
                    if (false)
#elif PREP_CLASS_BY_WAVE
                    /* the class chunks of a wave's OWN 64 rows (the rows whose {G, K, S, V} it has just stored):
                     * lanes 0..39 = 8 blocks x 5 chunks */
                    if (ER == PREP_THREADS) {
                        const int ln = tid & 63, wv = tid >> 6;
                        if (ln < 40) {
                            const int kbl = ln / 5, q = ln - kbl * 5, kb = ((row0 + 64 * wv) >> 3) + kbl;
                            if (kb * 8 <= H)
                                class_item(kb, q, [&](int vv, int qq, int4 x) {
                                    prep_store16(reinterpret_cast<int4*>(rcol + PREP_RROW(vv)) + qq, x);
                                });
                        }
                    } else
#endif
                    for (int it = tid; it < (ER / 8) * 5; it += PREP_THREADS) { /* class chunks: 1/8-resolution blocks x 5 */
                        const int kbl = it / 5, q = it - kbl * 5, kb = (row0 >> 3) + kbl;
                        if (kb * 8 <= H)
                            class_item(kb, q, [&](int v, int qq, int4 x) {
                                prep_store16(reinterpret_cast<int4*>(rcol + PREP_RROW(v)) + qq, x);
                            });
                    }
#ifndef PREP_ABL_WAVEROWS
                    if (r_lo >= row0 && r_lo < row0 + ER) instance_rows(true, rec_row);
#endif
                }
            }
        }
        /* (row H when it starts a block of its own -- H a multiple of PREP_EPI_ROWS: its class chunk and
         * {G, K, S, V} went out with that block above, its instance piece with row H - 1's owner) */
    } else if (PREP_STORE_LATE) {
        class_chunks();
        instance_rows(true, rec_row);
    }
}

__global__ __launch_bounds__(PREP_THREADS) void k_prepare_columns(
    const DevParams P, const float* __restrict__ joined, const int32_t* __restrict__ seg,
    const float* __restrict__ ground, const int* __restrict__ vhor_arr, RowRec* __restrict__ recs,
    int* __restrict__ col_flags, float* __restrict__ sv_arr, PruneRec* __restrict__ prune,
    int* __restrict__ n_generic) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    prepare_columns_body(P, (int)blockIdx.x, smem, joined, seg, ground, vhor_arr, recs, col_flags, sv_arr,
                         prune, n_generic);
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
/* The table is 8.6 GB for a batch of 64 (1024 x 2048 x 128): its stores ARE the kernel.  Measured (batch 64,
 * k_object_lut alone, tools/abl_prep.sh): 1.78 ms = 4.8 TB/s with plain stores; the same instruction stream
 * with the stores wrapped into 8 rows per column (absorbed by the L2, PREP_ABL_WRAP) 0.48 ms; non-temporal
 * stores (full 256-byte rows that nothing reads before the whole table is written: no reason to keep them in
 * the L2 / MALL) 1.59 ms = 5.4 TB/s.  Storing fewer fn per row only pays in whole 128-byte lines
 * (PREP_ABL_FNMAX=105: 2.37 ms, partial lines are read-modify-write). */
#ifndef PREP_LUT_NT
#define PREP_LUT_NT 1
#endif
/* carry_only (DevParams::lut_carry): only the rows 32 k -- the carries the 32-row blocks are chained through,
 * StixelsKernels.cu:268-272 -- are stored: 1/32 of the bytes.  The windowed unary ring kernel rebuilds the rows
 * it reads from them (the same network on the same values: bit-identical), generic columns get their whole table
 * from k_object_lut_generic afterwards. */
template <bool CARRY_ONLY>
__device__ __forceinline__ void object_lut_body(const DevParams& P, const int colg, const int fn_block,
                                                const int lane, const float* __restrict__ joined,
                                                const float* __restrict__ cost_T /*[dis][fn]*/,
                                                float* __restrict__ lutT) {
    const int H = P.H, D = P.D;
    const int fn = fn_block * 64 + lane;
#ifdef PREP_ABL_FNMAX /* ablation: how much of the LUT blocks' time is their store bytes */
    const bool fn_ok = fn < PREP_ABL_FNMAX;
    const int fnc = fn_ok ? fn : PREP_ABL_FNMAX - 1;
#else
    const bool fn_ok = fn < D;
    const int fnc = fn_ok ? fn : D - 1;
#endif
    const float* dcol = joined + (size_t)colg * H;
    float* lcol = lutT + (size_t)colg * (H + 1) * D;
    if (fn_ok) lcol[fn] = 0.0f; /* arr[0] = 0, :283-285 */
    float add = 0.0f;
    /* the per-row costs of the NEXT block are fetched before this block's 32 row stores are issued:
     * a wave's memory operations retire in order, so loads queued behind the stores would wait
     * for the whole store latency */
    float cn[LUT_BLOCK];
    /* (int)d of a block's 32 rows: one coalesced load, requested a block before its costs are (a
     * wave that is alone on its SIMD -- a single frame -- otherwise waits for two dependent memory
     * round trips per block: the disparities, then the table entries they select) */
    auto fetch_d = [&](int i) -> float {
        const int rl = i + (lane & (LUT_BLOCK - 1));
        return (rl < H) ? dcol[rl] : 0.0f; /* rows beyond the image use dis = 0, :244-247 */
    };
    auto fetch = [&](float d_l, float (&c)[LUT_BLOCK]) {
        int dis_l = (int)d_l;
        dis_l = min(max(dis_l, 0), D - 1); /* memory safety outside the input domain (Q8) */
#pragma unroll
        for (int l = 0; l < LUT_BLOCK; l++) { /* wave-uniform broadcasts */
            const int dis = __builtin_amdgcn_readlane(dis_l, l);
            c[l] = cost_T[(size_t)dis * D + fnc];
        }
    };
    fetch(fetch_d(0), cn);
    float d_next = fetch_d(LUT_BLOCK);
    /* One block: the next block's disparities and costs are requested, then this block's network runs
     * and its 32 rows are stored.  FULL blocks store unconditionally (lanes beyond D hold lane D - 1's
     * values and write them to its address again), so that the number of memory operations between
     * a request and its use is the same on every path and the compiler waits with a counted
     * s_waitcnt vmcnt(32) for the costs instead of vmcnt(0) for the previous block's stores as
     * well -- half of a block's time when the wave is alone on its SIMD (a single frame).  For the
     * same reason the first block is peeled: the loop is entered in its steady state. */
    auto block = [&](int i, bool full) {
        float c[LUT_BLOCK];
#pragma unroll
        for (int l = 0; l < LUT_BLOCK; l++) c[l] = cn[l];
        if (i + LUT_BLOCK < H) {
            const float d_here = d_next;
            d_next = fetch_d(i + 2 * LUT_BLOCK);
            fetch(d_here, cn);
        }
        c[0] += add; /* :249-251 */
#pragma unroll
        for (int j = 1; j < LUT_BLOCK; j <<= 1) { /* :255-263; descending l reads pre-step values */
#pragma unroll
            for (int l = LUT_BLOCK - 1; l >= j; l--) c[l] += c[l - j];
        }
        if (CARRY_ONLY) {
            if (fn_ok && i + LUT_BLOCK <= H) __builtin_nontemporal_store(c[LUT_BLOCK - 1], &lcol[(size_t)(i + LUT_BLOCK) * D + fn]);
        } else if (full) {
#pragma unroll
#ifdef PREP_ABL_WRAP /* ablation: the same stores into 8 rows per column (absorbed by the L2): what do the HBM bytes cost */
            for (int l = 0; l < LUT_BLOCK; l++) lcol[(size_t)((i + l + 1) & 7) * D + fnc] = c[l];
#elif PREP_LUT_NT
            for (int l = 0; l < LUT_BLOCK; l++) __builtin_nontemporal_store(c[l], &lcol[(size_t)(i + l + 1) * D + fnc]);
#else
            for (int l = 0; l < LUT_BLOCK; l++) lcol[(size_t)(i + l + 1) * D + fnc] = c[l]; /* :266 */
#endif
        } else if (fn_ok) {
#pragma unroll
            for (int l = 0; l < LUT_BLOCK; l++)
                if (i + l < H) lcol[(size_t)(i + l + 1) * D + fn] = c[l];
        }
        add = c[LUT_BLOCK - 1]; /* :268-272 */
    };
    int i = 0;
    if (LUT_BLOCK <= H) { /* peeled */
        block(0, true);
        i = LUT_BLOCK;
    }
    for (; i + LUT_BLOCK <= H; i += LUT_BLOCK) block(i, true);
    if (i < H) block(i, false);
}

__global__ __launch_bounds__(64) void k_object_lut(const DevParams P,
                                                   const float* __restrict__ joined,
                                                   const float* __restrict__ cost_T,
                                                   float* __restrict__ lutT) {
    object_lut_body<false>(P, (int)blockIdx.x, (int)blockIdx.y, (int)threadIdx.x, joined, cost_T, lutT);
}

/* Behind a fused LUT + DP launch (k_dp_unary_fast, LUTF) whose workgroups could not trust the hand-over: the complete
 * table again, by the ordinary units.  Leaves at once while the word is 0 -- the normal case. */
__global__ __launch_bounds__(64) void k_object_lut_repair(const DevParams P, int ncols, const float* __restrict__ joined,
                                                          const float* __restrict__ cost_T, float* __restrict__ lutT,
                                                          const int* __restrict__ run_if) {
    if (__builtin_amdgcn_readfirstlane(*run_if) == 0) return;
    const int fn_blocks = (P.D + 63) / 64;
    for (int u = (int)blockIdx.x; u < ncols * fn_blocks; u += (int)gridDim.x) {
        const int colg = u / fn_blocks;
        object_lut_body<false>(P, colg, u - colg * fn_blocks, (int)threadIdx.x, joined, cost_T, lutT);
    }
}

/* After a carry-only prepare: the complete table of the GENERIC columns (k_dp_unary reads it as it is).  A small
 * grid that leaves at once when the prepare kernel counted no generic column -- the normal case. */
__global__ __launch_bounds__(64) void k_object_lut_generic(const DevParams P, int ncols, const float* __restrict__ joined,
                                                           const float* __restrict__ cost_T, float* __restrict__ lutT,
                                                           const int* __restrict__ col_flags,
                                                           const int* __restrict__ n_generic) {
    if (__builtin_amdgcn_readfirstlane(*n_generic) == 0) return;
    const int fn_blocks = (P.D + 63) / 64;
    for (int u = (int)blockIdx.x; u < ncols * fn_blocks; u += (int)gridDim.x) {
        const int colg = u / fn_blocks;
        if (__builtin_amdgcn_readfirstlane(col_flags[colg]) == 0) continue;
        object_lut_body<false>(P, colg, u - colg * fn_blocks, (int)threadIdx.x, joined, cost_T, lutT);
    }
}

/* Both preparation kernels in ONE launch: a 256-thread workgroup is either one column of
 * k_prepare_columns (blocks 0 .. ncols - 1) or four (column, 64 fn) units of k_object_lut (the
 * blocks after them).  A frame or a few: neither kernel fills the chip and both are latency chains
 * (59 + 45 us one after the other, 51 us together; on two streams they did not overlap in
 * practice).  A batch of 64: the LUT blocks start while the last column blocks drain, 2.85 instead
 * of 3.1 ms for two launches.  MEASURED alternatives: the two kinds interleaved in block order, so
 * that they share the CUs for the whole launch: 3.7 ms (and a single frame 0.349 instead of
 * 0.325 ms) -- side by side they take the memory system from each other.  (Until the LUT loop
 * stored unconditionally its body took 166 VGPRs under this kernel's launch bounds and the fused
 * launch cost a large batch the occupancy both bodies live on: 8.2 ms.) */
#ifndef IS_FUSED_LUT_FIRST
#define IS_FUSED_LUT_FIRST 0
#endif
#ifndef IS_FUSED_INTERLEAVE
#define IS_FUSED_INTERLEAVE 0
#endif
__global__ __launch_bounds__(PREP_THREADS) void k_prepare_fused(
    const DevParams P, int ncols, int n_lut, const float* __restrict__ joined, const int32_t* __restrict__ seg,
    const float* __restrict__ ground, const int* __restrict__ vhor_arr, const float* __restrict__ cost_T,
    RowRec* __restrict__ recs, float* __restrict__ lutT, int* __restrict__ col_flags,
    float* __restrict__ sv_arr, PruneRec* __restrict__ prune, int* __restrict__ n_generic) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = (int)blockIdx.x;
    bool is_lut;
    int lut_b, col_b;
#if IS_FUSED_INTERLEAVE
    /* block order: IS_FUSED_INTERLEAVE column blocks, one LUT block, ... while both kinds last */
    {
        constexpr int G = IS_FUSED_INTERLEAVE + 1;
        const int groups = min(ncols / IS_FUSED_INTERLEAVE, n_lut); /* full groups */
        if (b < groups * G) {
            const int g = b / G, r = b - g * G;
            is_lut = r == IS_FUSED_INTERLEAVE;
            lut_b = g;
            col_b = g * IS_FUSED_INTERLEAVE + r;
        } else { /* the rest of the longer kind */
            const int rest = b - groups * G;
            const int cols_left = ncols - groups * IS_FUSED_INTERLEAVE;
            is_lut = rest >= cols_left;
            col_b = groups * IS_FUSED_INTERLEAVE + rest;
            lut_b = groups + (rest - cols_left);
        }
    }
#else
    is_lut = IS_FUSED_LUT_FIRST ? b < n_lut : b >= ncols;
    lut_b = IS_FUSED_LUT_FIRST ? b : b - ncols;
    col_b = IS_FUSED_LUT_FIRST ? b - n_lut : b;
#endif
    if (is_lut) {
        const int fn_blocks = (P.D + 63) / 64;
        const int unit = lut_b * (PREP_THREADS / 64) + (int)(threadIdx.x >> 6);
        if (unit < ncols * fn_blocks) {
            if (P.lut_carry)
                object_lut_body<true>(P, unit / fn_blocks, unit % fn_blocks, (int)(threadIdx.x & 63), joined, cost_T, lutT);
            else
                object_lut_body<false>(P, unit / fn_blocks, unit % fn_blocks, (int)(threadIdx.x & 63), joined, cost_T, lutT);
        }
    } else {
        prepare_columns_body(P, col_b, smem, joined, seg, ground, vhor_arr, recs,
                             col_flags, sv_arr, prune, n_generic);
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

extern "C" {

size_t isk_prepare_lds_bytes(const DevParams* P) {
    return sizeof(float) * (size_t)prep_scan_leaves(P->H, P->P2) * 3 +
           sizeof(int32_t) * (size_t)P->CH * prep_seg_stride(P->H) + 192
#ifdef PREP_LDS_PAD /* occupancy experiments */
           + PREP_LDS_PAD
#endif
        ;
}

hipError_t isk_launch_lut_repair(const DevParams* P, int ncols, const float* joined, const float* cost_T, float* lutT,
                                 hipStream_t stream) {
    const int units = ncols * ((P->D + 63) / 64);
    hipLaunchKernelGGL(k_object_lut_repair, dim3(units < 8192 ? units : 8192), dim3(64), 0, stream, *P, ncols, joined,
                       cost_T, lutT, P->lutf_bad);
    return hipGetLastError();
}

hipError_t isk_launch_prepare(const DevParams* P, int ncols, const float* joined,
                              const int32_t* seg, const float* ground, const int* vhor,
                              const float* cost_T, RowRec* recs, float* lutT,
                              int* col_flags, float* sv_arr, PruneRec* prune, int* n_generic,
                              hipStream_t stream,
                              hipStream_t aux, hipEvent_t ev_fork, hipEvent_t ev_join) {
    /* The two prepare kernels are independent: one launch with workgroups of both kinds
     * (k_prepare_fused) by default.  IS_PREPARE_OVERLAP = 0: two launches in order on one stream,
     * 1: two launches on two streams (did not overlap in practice), 2: the default. */
    const bool side_by_side = aux != nullptr && P->knob_prepare_overlap == 1;
    hipError_t e;
    if (P->lut_fused) { /* the LUT units run inside the unary DP launch (k_dp_unary_fast, LUTF): records only here */
        hipLaunchKernelGGL(k_prepare_columns, dim3(ncols), dim3(PREP_THREADS), isk_prepare_lds_bytes(P), stream, *P,
                           joined, seg, ground, vhor, recs, col_flags, sv_arr, prune, n_generic);
        return hipGetLastError();
    }
    const bool fused = P->knob_prepare_overlap == 2 || P->knob_prepare_overlap < 0;
    if (fused) {
        const int units = ncols * ((P->D + 63) / 64);
        const int n_lut = (units + PREP_THREADS / 64 - 1) / (PREP_THREADS / 64);
        hipLaunchKernelGGL(k_prepare_fused, dim3(ncols + n_lut), dim3(PREP_THREADS),
                           isk_prepare_lds_bytes(P), stream, *P, ncols, n_lut, joined, seg, ground, vhor,
                           cost_T, recs, lutT, col_flags, sv_arr, prune, n_generic);
        if (P->lut_carry) {
            const int g = units < 2048 ? units : 2048;
            hipLaunchKernelGGL(k_object_lut_generic, dim3(g), dim3(64), 0, stream, *P, ncols, joined, cost_T, lutT,
                               col_flags, n_generic);
        }
        return hipGetLastError();
    }
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
                       col_flags, sv_arr, prune, n_generic);
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

hipError_t isk_set_lds_prepare(const DevParams* P) {
    hipError_t e = hipFuncSetAttribute((const void*)k_prepare_columns, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)isk_prepare_lds_bytes(P));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)k_prepare_fused, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)isk_prepare_lds_bytes(P));
}

} /* extern "C" */
