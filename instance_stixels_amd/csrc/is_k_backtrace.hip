/* is_k_backtrace.hip -- back-tracing of the DP tables into Sections, instance candidates.
 * See is_kernels.h. */
#include "is_kernels.h"

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
 * (StixelsKernels.cu:843-955); only the index chase is inherently serial, so: lane 0 walks the
 * chain and records the cuts, then the lanes build the Sections in parallel (one per lane).  The
 * chase reads the tables where they lie (two dependent loads per section, a few dozen sections):
 * staging the column's 24 KB of tables in LDS first (round 1) limited a CU to five waves. */
/* STAGE (few columns, i.e. latency matters more than waves per CU): the column's two tables are
 * copied into LDS first (coalesced, one round trip) and the chase runs there (~0.1 us per section
 * instead of a dependent global round trip): one frame 28 -> ~8 us. */
/* TWO (large batches): two columns per wavefront -- lanes 0 and 32 chase their chains side by side, each half
 * builds its column's Sections with its 32 lanes.  A CU holds 32 waves, the chip 8192: 16384 one-column waves
 * are two rounds of a latency chain (one dependent memory round trip per section), 8192 two-column waves one. */
template <bool STAGE, bool TWO = false>
__global__ __launch_bounds__(64) void k_backtrace(const DevParams P, int ncols, int pairwise,
                                                  const RowRec* __restrict__ recs,
                                                  const float* __restrict__ cost_table,
                                                  const int32_t* __restrict__ index_table,
                                                  const int* __restrict__ col_flags,
                                                  is_section* __restrict__ sections,
                                                  int* __restrict__ inst_cnt /* [ncols][8] or null */,
                                                  int* __restrict__ n_generic /* reset for the next call */) {
    static_assert(!(STAGE && TWO), "the staged variant is for small calls: one column per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    constexpr int LW = TWO ? 32 : 64;          /* lanes per column */
    const int half = TWO ? (lane >> 5) : 0, li = lane & (LW - 1);
    const int colq = TWO ? (int)blockIdx.x * 2 + half : (int)blockIdx.x;
    if ((TWO ? (int)blockIdx.x * 2 : (int)blockIdx.x) >= ncols) return;
    const bool valid = colq < ncols;           /* (TWO, odd column count: the last wave's second half idles) */
    const int colg = valid ? colq : ncols - 1;
    /* the count of generic-encoding columns (k_prepare_columns adds to it, the generic DP kernels of
     * this call have read it: they precede this launch in stream order) goes back to zero here --
     * a memset node per call costs a single frame ten microseconds of queue time */
    if (blockIdx.x == 0 && lane == 0) *n_generic = 0;
    const int H = P.H, S = P.S;
    int* s_cut = (int*)smem + half * (3 * S + 8); /* [S][3]: vT, vB, type */
    int* s_n = s_cut + 3 * S;                     /* [1] */
    const bool wide = col_flags[colg] != 0; /* generic record encoding, see RowRec */
    const RowRec* rcol = recs + (size_t)colg * (H + 1);
    const float* ct = cost_table + (size_t)colg * H * 3;
    const int32_t* it = index_table + (size_t)colg * H * 3;
    is_section* out = sections + (size_t)colg * S;
    const float* s_cost = ct; /* (not STAGE: the walk below reads global memory) */
    const int32_t* s_idx = it;
    if (STAGE) {
        float* l_cost = (float*)(s_n + 4);
        int32_t* l_idx = (int32_t*)(l_cost + 3 * H);
        for (int i = lane; i < 3 * H; i += 64) {
            l_cost[i] = ct[i];
            l_idx[i] = it[i];
        }
        __syncthreads();
        s_cost = l_cost;
        s_idx = l_idx;
    }
    if (li == 0) {
        int vT = H - 1;
        const float last_ground = s_cost[vT * 3 + IS_GROUND];
        const float last_object = s_cost[vT * 3 + IS_OBJECT];
        const float last_sky = s_cost[vT * 3 + IS_SKY];
        int type = IS_OBJECT; /* :854-861 */
        if (last_ground < last_object) type = IS_GROUND;
        if (last_sky < __builtin_fminf(last_ground, last_object)) type = IS_SKY;
        int n = 0;
        int prev_vT;
        /* the three index entries of the current row travel with the walk: a hop then costs ONE
         * dependent memory round trip (the costs and the indices of row vB - 1 together), not two */
        int i0 = s_idx[vT * 3 + 0], i1 = s_idx[vT * 3 + 1], i2 = s_idx[vT * 3 + 2];
        do {
            const int raw = type == 0 ? i0 : (type == 1 ? i1 : i2);
            int vB, prev_type;
            if (pairwise) {
                vB = raw / 3;
                prev_type = raw % 3;
                if (vB > 0) {
                    i0 = s_idx[(vB - 1) * 3 + 0]; i1 = s_idx[(vB - 1) * 3 + 1]; i2 = s_idx[(vB - 1) * 3 + 2];
                }
            } else {
                /* unary: index_table holds the winning vB; the predecessor type is the arg-min
                 * of the FINAL cost_table[vB-1], tie rules of :723-727, 769-773, 828-835 */
                vB = raw;
                prev_type = IS_OBJECT;
                if (vB > 0) {
                    const float cG = s_cost[(vB - 1) * 3 + IS_GROUND];
                    const float cO = s_cost[(vB - 1) * 3 + IS_OBJECT];
                    const float cS = s_cost[(vB - 1) * 3 + IS_SKY];
                    i0 = s_idx[(vB - 1) * 3 + 0]; i1 = s_idx[(vB - 1) * 3 + 1]; i2 = s_idx[(vB - 1) * 3 + 2];
                    if (cG < cO) prev_type = IS_GROUND;
                    if (type == IS_OBJECT && cS < __builtin_fminf(cG, cO)) prev_type = IS_SKY;
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
    /* (wave-uniform trip count: ballots inside; TWO: the longer of the two columns) */
    const int n_loop = TWO ? max(__builtin_amdgcn_readlane(n, 0), __builtin_amdgcn_readlane(n, 32)) : n;
    const unsigned long long half_mask = TWO ? (half ? 0xFFFFFFFF00000000ull : 0x00000000FFFFFFFFull) : ~0ull;
    int my_cnt = 0; /* lane k < 8 (of the column's lanes): instance candidates of class 11 + k in this column */
    for (int i0 = 0; i0 <= n_loop; i0 += LW) {
        const int i = i0 + li;
        is_section sec;
        sec.type = -1; sec.vB = 0; sec.vT = 0; sec.disparity = 0.0f; /* terminator, :952-954 */
        sec.semantic_class = 0; sec.cost = 0.0f; sec.instance_meanx = 0.0f; sec.instance_meany = 0.0f;
        if (i < n) {
            const int vT = s_cut[i * 3 + 0], vB = s_cut[i * 3 + 1], type = s_cut[i * 3 + 2];
            sec = make_section(P, rcol, wide, vT, vB, type, s_cost[vT * 3 + type]);
        }
        if (i <= n && valid) out[i] = sec;
        if (inst_cnt) { /* candidates per instance class, :926-942 (the scatter: k_compact_instances) */
            const bool cand = i < n && sec.type == IS_OBJECT && sec.semantic_class >= IS_FIRST_INSTANCE_CLASS;
            const int k = sec.semantic_class - IS_FIRST_INSTANCE_CLASS;
#pragma unroll
            for (int kk = 0; kk < IS_INSTANCE_CLASSES; kk++) {
                const int c = __builtin_popcountll(__builtin_amdgcn_ballot_w64(cand && k == kk) & half_mask);
                if (li == kk) my_cnt += c;
            }
        }
    }
    if (inst_cnt && li < IS_INSTANCE_CLASSES && valid) inst_cnt[(size_t)colg * IS_INSTANCE_CLASSES + li] = my_cnt;
}

/* ====================================================================================== */
/* Instance candidates in canonical (column, section) order, reference layout              */
/* (StixelsKernels.cu:926-942), the whole batch in ONE launch                              */
/* ====================================================================================== */
/* grid = (ISC_CHUNKS, n_images).  k_backtrace has left the per-column, per-class candidate counts
 * in inst_cnt; every workgroup turns its image's C x 8 counts into exclusive offsets (a few
 * thousand integers: cheaper to repeat per chunk than to hand over through memory), then its
 * waves scatter the candidates of the chunk's columns, one wave per column at a time: lane i
 * reads Section i, its slot is offset[column][class] + the number of same-class candidates in
 * the lanes below it (ballot + mbcnt).  `tbl[image]`: the caller's per-image output arrays. */
#define ISC_THREADS 256
#define ISC_CHUNKS 16
__global__ __launch_bounds__(ISC_THREADS) void k_compact_instances(
    const DevParams P, const is_section* __restrict__ sections, const int* __restrict__ inst_cnt,
    const is_instance_buffers* __restrict__ tbl) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* s_off = (int*)smem;                    /* [C][8] exclusive offsets */
    int* s_part = s_off + (size_t)P.C * IS_INSTANCE_CLASSES; /* [ISC_THREADS / 8][8] */
    const int C = P.C, S = P.S;
    const int img = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const is_instance_buffers ib = tbl[img];
    if (!ib.d_centerofmass && !ib.d_indices && !ib.d_core_candidates && !ib.d_instances_per_class)
        return; /* nothing wanted for this image */
    const is_section* sec = sections + (size_t)img * C * S;
    const int* cnt = inst_cnt + (size_t)img * C * IS_INSTANCE_CLASSES;
    /* exclusive scan over the columns, per class: thread (g, k) sums the columns of group g,
     * a serial pass over the ISC_THREADS / 8 group totals, then the groups are written out */
    constexpr int NG = ISC_THREADS / IS_INSTANCE_CLASSES;
    const int k = tid & (IS_INSTANCE_CLASSES - 1), g = tid / IS_INSTANCE_CLASSES;
    const int per = (C + NG - 1) / NG;
    const int c_lo = min(g * per, C), c_hi = min(c_lo + per, C);
    int sum = 0;
    for (int c = c_lo; c < c_hi; c++) sum += cnt[c * IS_INSTANCE_CLASSES + k];
    s_part[g * IS_INSTANCE_CLASSES + k] = sum;
    __syncthreads();
    int base = 0;
    for (int gg = 0; gg < g; gg++) base += s_part[gg * IS_INSTANCE_CLASSES + k];
    if (g == NG - 1 && blockIdx.x == 0 && ib.d_instances_per_class) ib.d_instances_per_class[k] = base + sum;
    for (int c = c_lo; c < c_hi; c++) {
        s_off[c * IS_INSTANCE_CLASSES + k] = base;
        base += cnt[c * IS_INSTANCE_CLASSES + k];
    }
    __syncthreads();
    /* scatter: the chunk's columns, one wave per column */
    const int cols_per_chunk = (C + ISC_CHUNKS - 1) / ISC_CHUNKS;
    const int cc_lo = blockIdx.x * cols_per_chunk, cc_hi = min(cc_lo + cols_per_chunk, C);
    for (int c = cc_lo + w; c < cc_hi; c += ISC_THREADS / 64) {
        int run[IS_INSTANCE_CLASSES];
#pragma unroll
        for (int kk = 0; kk < IS_INSTANCE_CLASSES; kk++) run[kk] = s_off[c * IS_INSTANCE_CLASSES + kk];
        for (int i0 = 0; i0 < S; i0 += 64) { /* wave-uniform trip count */
            const int i = i0 + lane;
            is_section s;
            s.type = -1;
            if (i < S) s = sec[(size_t)c * S + i];
            /* everything behind the terminator is unspecified: cut at the first -1 */
            const unsigned long long term = __builtin_amdgcn_ballot_w64(s.type == -1);
            const unsigned long long below_term = term ? ((term & (0 - term)) - 1ull) : ~0ull;
            const bool valid = (below_term >> lane) & 1ull;
            const bool cand = valid && s.type == IS_OBJECT && s.semantic_class >= IS_FIRST_INSTANCE_CLASS;
            const int kc = s.semantic_class - IS_FIRST_INSTANCE_CLASS;
            int slot = -1;
#pragma unroll
            for (int kk = 0; kk < IS_INSTANCE_CLASSES; kk++) {
                const unsigned long long m = __builtin_amdgcn_ballot_w64(cand && kc == kk);
                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                           __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                if (cand && kc == kk) slot = run[kk] + rank;
                run[kk] += __builtin_popcountll(m);
            }
            if (cand) {
                const size_t o = (size_t)kc * C * S + slot;
                if (ib.d_centerofmass) {
                    ib.d_centerofmass[o * 2] = s.instance_meanx;
                    ib.d_centerofmass[o * 2 + 1] = s.instance_meany;
                }
                if (ib.d_indices) { ib.d_indices[o * 2] = c; ib.d_indices[o * 2 + 1] = i; }
                if (ib.d_core_candidates) ib.d_core_candidates[o] = (s.vT + 1 - s.vB) >= P.size_filter;
            }
            if (term) break;
        }
    }
}

extern "C" {

#ifndef IS_BACKTRACE_TWO_MIN_COLS
#define IS_BACKTRACE_TWO_MIN_COLS 8192 /* more one-column waves than the chip holds at once */
#endif
hipError_t isk_launch_backtrace(const DevParams* P, int ncols, int pairwise, const RowRec* recs,
                                const float* cost_table, const int32_t* index_table,
                                const int* col_flags, is_section* sections, int* inst_cnt,
                                int* n_generic, hipStream_t stream) {
    const size_t lds = sizeof(int) * (3 * (size_t)P->S + 8);
    const size_t lds_staged = lds + sizeof(int) * 6 * (size_t)P->H;
    if (ncols <= IS_BACKTRACE_STAGE_MAX_COLS && lds_staged <= 64 * 1024)
        hipLaunchKernelGGL(k_backtrace<true>, dim3(ncols), dim3(64), lds_staged, stream, *P, ncols, pairwise,
                           recs, cost_table, index_table, col_flags, sections, inst_cnt, n_generic);
    else if (ncols >= IS_BACKTRACE_TWO_MIN_COLS)
        hipLaunchKernelGGL((k_backtrace<false, true>), dim3((ncols + 1) / 2), dim3(64), 2 * lds, stream, *P, ncols,
                           pairwise, recs, cost_table, index_table, col_flags, sections, inst_cnt, n_generic);
    else
        hipLaunchKernelGGL(k_backtrace<false>, dim3(ncols), dim3(64), lds, stream, *P, ncols, pairwise, recs,
                           cost_table, index_table, col_flags, sections, inst_cnt, n_generic);
    return hipGetLastError();
}

static size_t compact_lds_bytes(const DevParams* P) {
    return sizeof(int) * ((size_t)P->C * IS_INSTANCE_CLASSES + ISC_THREADS) + 16;
}

/* the instance candidates of `n_images` images: one launch */
hipError_t isk_launch_compact(const DevParams* P, int n_images, const is_section* sections,
                              const int* inst_cnt, const is_instance_buffers* d_tbl,
                              hipStream_t stream) {
    hipLaunchKernelGGL(k_compact_instances, dim3(ISC_CHUNKS, n_images), dim3(ISC_THREADS),
                       compact_lds_bytes(P), stream, *P, sections, inst_cnt, d_tbl);
    return hipGetLastError();
}

hipError_t isk_set_lds_backtrace(const DevParams* P) {
    hipError_t e = hipFuncSetAttribute((const void*)k_backtrace<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(sizeof(int) * (3 * (size_t)P->S + 8)));
    if (e != hipSuccess) return e;
    if (sizeof(int) * (3 * (size_t)P->S + 8 + 6 * (size_t)P->H) <= 64 * 1024) {
        e = hipFuncSetAttribute((const void*)k_backtrace<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(sizeof(int) * (3 * (size_t)P->S + 8 + 6 * (size_t)P->H)));
        if (e != hipSuccess) return e;
    }
    /* more than ~2000 stixel columns: the per-column offsets exceed the 64 KiB default */
    return hipFuncSetAttribute((const void*)k_compact_instances, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)compact_lds_bytes(P));
}

} /* extern "C" */
