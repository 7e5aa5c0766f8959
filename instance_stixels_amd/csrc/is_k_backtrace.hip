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
    int* s_cut = (int*)smem;                /* [S][3]: vT, vB, type */
    int* s_n = s_cut + 3 * S;               /* [1] */
    const bool wide = col_flags[colg] != 0; /* generic record encoding, see RowRec */
    const RowRec* rcol = recs + (size_t)colg * (H + 1);
    const float* ct = cost_table + (size_t)colg * H * 3;
    const int32_t* it = index_table + (size_t)colg * H * 3;
    is_section* out = sections + (size_t)colg * S;
    const float* s_cost = ct; /* (names kept: the walk below reads global memory) */
    const int32_t* s_idx = it;
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

extern "C" {

hipError_t isk_launch_backtrace(const DevParams* P, int ncols, int pairwise, const RowRec* recs,
                                const float* cost_table, const int32_t* index_table,
                                const int* col_flags, is_section* sections, hipStream_t stream) {
    const size_t lds = sizeof(int) * (3 * (size_t)P->S + 4);
    hipLaunchKernelGGL(k_backtrace, dim3(ncols), dim3(64), lds, stream, *P, ncols, pairwise, recs,
                       cost_table, index_table, col_flags, sections);
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

hipError_t isk_set_lds_backtrace(const DevParams* P) {
    hipError_t e = hipFuncSetAttribute((const void*)k_backtrace, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(sizeof(int) * (3 * (size_t)P->S + 4)));
    if (e != hipSuccess) return e;
    /* more than 2047 stixel columns: the per-column counters exceed the 64 KiB default */
    return hipFuncSetAttribute((const void*)k_compact_instances, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(sizeof(int) * (size_t)P->C * IS_INSTANCE_CLASSES + 16));
}

} /* extern "C" */
