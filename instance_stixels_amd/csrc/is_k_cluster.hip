/* is_k_cluster.hip -- size-filtered DBSCAN over the predicted instance centres.
 *
 * The reference calls `ML::dbscanFit` of a cuML fork whose source is not in its tree
 * (/root/reference/InstanceStixels/src/Stixels.cu:639-681; branch `dbscan-sizefilter`, no commit
 * pinned, singularity_recipe:108-110).  The semantics built here are those of its Python twin
 * (/root/reference/tools/visualization/clustering_visualization.py:894-960):
 *
 *   large = candidates with height >= size_filter (the core-candidate flag written by the
 *           back-trace, StixelsKernels.cu:940-941); nothing is labelled unless
 *           #large > min_pts (:926);
 *   DBSCAN(eps, min_samples = min_pts) over the large points only (:928-929): a large point is a
 *           core point when >= min_pts large points (itself included) lie within eps; core
 *           points within eps of each other share a cluster; clusters are numbered in the order
 *           of their first core point; a large non-core point takes the cluster that reaches it
 *           first (= the lowest-numbered one among its core neighbours) or -1;
 *   every small point takes the label of its NEAREST core point if that lies within eps, else
 *           -1 (:935-948; first core on ties).
 *
 * Distances: fp32, dx*dx + dy*dy compared with eps*eps (no contraction), like cuML on float
 * input.  One workgroup per (image, instance class); N <= realcols * max_sections, in practice a
 * few hundred, so the O(N^2) neighbour sweeps are a few microseconds and nothing leaves the
 * device: Stixels::Compute no longer copies candidates to the host to cluster them.
 * A batch is ONE launch: grid = (8 classes, n_images).
 */
#include "is_kernels.h"

#define CLU_THREADS 256

/* LDS pointer types: with them the sweeps compile to ds_read (pipelined, unrolled) instead of flat loads */
typedef float clu_f2 __attribute__((ext_vector_type(2))); /* (a builtin vector: loadable from any address space) */
typedef __attribute__((address_space(3))) clu_f2 lds_float2;
typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) int32_t lds_i32;

__device__ __forceinline__ float clu_d2(const clu_f2 a, const clu_f2 b) {
    const float dx = a.x - b.x, dy = a.y - b.y;
    return dx * dx + dy * dy;
}

/* labels doubles as the component array while the kernel runs:
 *   >= 0  core point, value = smallest core index known to be in the same cluster
 *   -2    large, not core        -3   small
 * rank / out are scratch of n ints each. */
#define CLU_LDS_N 2048 /* classes with up to this many candidates are clustered out of LDS copies */
/* COM / CAND / LAB: pointers to the centre, large-flag and component arrays -- global memory, or the
 * LDS copies (address_space(3) types, so that the sweeps are ds_read code the compiler can unroll and
 * pipeline).  labels_out: the caller's array in global memory. */
template <class COM, class CAND, class LAB>
__device__ __forceinline__ void cluster_body(int n, float eps2, int min_pts, COM com, CAND cand, LAB labels,
                                             int32_t* const labels_out, int32_t* rank, int32_t* out,
                                             int* s_red) {
    const int tid = threadIdx.x;
    /* number of large points: the twin clusters only if it exceeds min_pts (:926) */
    int cnt = 0;
    for (int i = tid; i < n; i += CLU_THREADS) cnt += cand[i] != 0;
    s_red[tid] = cnt;
    __syncthreads();
    for (int s = CLU_THREADS / 2; s > 0; s >>= 1) {
        if (tid < s) s_red[tid] += s_red[tid + s];
        __syncthreads();
    }
    const int n_large = s_red[0];
    __syncthreads();
    if (n_large <= min_pts) {
        for (int i = tid; i < n; i += CLU_THREADS) labels_out[i] = -1;
        return;
    }

    /* core points */
    for (int i = tid; i < n; i += CLU_THREADS) {
        int l = -3;
        if (cand[i]) {
            const clu_f2 p = com[i];
            int c = 0;
            for (int j = 0; j < n; j++) c += (cand[j] != 0) && (clu_d2(p, com[j]) <= eps2);
            l = (c >= min_pts) ? i : -2;
        }
        labels[i] = l;
    }
    __syncthreads();

    /* connected components of the core points: minimum-index propagation with pointer jumping;
     * labels only ever decrease, so reading a neighbour's value mid-update is harmless */
    for (;;) {
        int changed = 0;
        for (int i = tid; i < n; i += CLU_THREADS) {
            const int li = labels[i];
            if (li < 0) continue;
            const clu_f2 p = com[i];
            int m = li;
            for (int j = 0; j < n; j++) {
                const int lj = labels[j];
                if (lj >= 0 && lj < m && clu_d2(p, com[j]) <= eps2) m = lj;
            }
            while (labels[m] < m) m = labels[m]; /* jump to the current root */
            if (m < li) { labels[i] = m; changed = 1; }
        }
        if (!__syncthreads_or(changed)) break;
    }

    /* cluster number = rank of the root (smallest core index of the cluster) among the roots:
     * the order in which a scan over the points discovers the clusters */
    {
        const int per = (n + CLU_THREADS - 1) / CLU_THREADS;
        const int lo = tid * per, hi = min(lo + per, n);
        int c = 0;
        for (int i = lo; i < hi; i++) c += labels[i] == i;
        s_red[tid] = c;
        __syncthreads();
        int base = 0;
        for (int t = 0; t < tid; t++) base += s_red[t];
        for (int i = lo; i < hi; i++) {
            rank[i] = base;
            base += labels[i] == i;
        }
    }
    __syncthreads();

    for (int i = tid; i < n; i += CLU_THREADS) {
        const int li = labels[i];
        int res = -1;
        if (li >= 0) {
            res = rank[li];
        } else {
            const clu_f2 p = com[i];
            if (li == -2) { /* border point: lowest-numbered cluster among the core neighbours */
                int best = n;
                for (int j = 0; j < n; j++) {
                    const int lj = labels[j];
                    if (lj >= 0 && lj < best && clu_d2(p, com[j]) <= eps2) best = lj;
                }
                if (best < n) res = rank[best];
            } else { /* small point: nearest core point, first one on ties, within eps */
                float bd = IS_INF;
                int bj = -1;
                for (int j = 0; j < n; j++) {
                    if (labels[j] < 0) continue;
                    const float d = clu_d2(p, com[j]);
                    if (d < bd) { bd = d; bj = j; }
                }
                if (bj >= 0 && bd <= eps2) res = rank[labels[bj]];
            }
        }
        out[i] = res;
    }
    __syncthreads();
    for (int i = tid; i < n; i += CLU_THREADS) labels_out[i] = out[i];
}


__device__ __forceinline__ void cluster_class(int n, float eps2, int min_pts, const clu_f2* com,
                                              const uint8_t* cand, int32_t* const labels_out, int32_t* rank,
                                              int32_t* out, int* s_red, clu_f2* s_com, uint8_t* s_cand,
                                              int32_t* s_lab) {
    const int tid = threadIdx.x;
    if (n == 0) return;
    /* The sweeps read centre, flag and component of EVERY candidate j for every candidate i: the
     * same address in all lanes, one dependent L2 round trip per j when the arrays lie in global
     * memory (a class of 300 candidates: 85 us).  Up to CLU_LDS_N candidates the three arrays are
     * copied into LDS first. */
    if (n <= CLU_LDS_N) {
        for (int i = tid; i < n; i += CLU_THREADS) { s_com[i] = com[i]; s_cand[i] = cand[i]; }
        __syncthreads();
        cluster_body(n, eps2, min_pts, (const lds_float2*)s_com, (const lds_u8*)s_cand, (lds_i32*)s_lab,
                     labels_out, rank, out, s_red);
    } else {
        cluster_body(n, eps2, min_pts, com, cand, labels_out, labels_out, rank, out, s_red);
    }
}

/* One workgroup per (instance class, image): grid = (8, n_images).  `tbl[image]` holds the
 * image's candidate arrays (NULL: the single image `one`); images without d_labels are skipped.
 * `packed` (optional, per image): packed[0] = number of candidates of all classes, then one
 * (column, section index, label) triple per candidate, classes in ascending order -- everything
 * Stixels::GetInstanceStixels needs in one small copy.  scratch: [n_images][classes][2][n_slots]. */
__global__ __launch_bounds__(CLU_THREADS) void k_cluster_instances(
    int n_slots, float eps2, int min_pts, const is_instance_buffers* __restrict__ tbl,
    const is_instance_buffers one, int32_t* __restrict__ scratch) {
    __shared__ int s_red[CLU_THREADS];
    __shared__ clu_f2 s_com[CLU_LDS_N];
    __shared__ int32_t s_lab[CLU_LDS_N];
    __shared__ uint8_t s_cand[CLU_LDS_N];
    const int cls = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
    const is_instance_buffers ib = tbl ? tbl[img] : one;
    if (!ib.d_labels) return;
    const int32_t* per_class = ib.d_instances_per_class;
    const int n = min(max(per_class[cls], 0), n_slots);
    const clu_f2* com = reinterpret_cast<const clu_f2*>(ib.d_centerofmass) + (size_t)cls * n_slots;
    const uint8_t* cand = ib.d_core_candidates + (size_t)cls * n_slots;
    int32_t* labels = ib.d_labels + (size_t)cls * n_slots;
    int32_t* rank = scratch + ((size_t)img * IS_INSTANCE_CLASSES + cls) * 2 * n_slots;
    cluster_class(n, eps2, min_pts, com, cand, labels, rank, rank + n_slots, s_red, s_com, s_cand, s_lab);
    int32_t* packed = ib.d_packed;
    if (packed && ib.d_indices) {
        __syncthreads();
        int base = 0, total = 0;
        for (int k = 0; k < IS_INSTANCE_CLASSES; k++) {
            const int m = min(max(per_class[k], 0), n_slots);
            if (k < cls) base += m;
            total += m;
        }
        if (cls == 0 && tid == 0) packed[0] = total;
        const int32_t* idx = ib.d_indices + (size_t)cls * n_slots * 2;
        for (int i = tid; i < n; i += CLU_THREADS) {
            int32_t* t = packed + 1 + (size_t)(base + i) * 3;
            t[0] = idx[2 * i]; t[1] = idx[2 * i + 1]; t[2] = labels[i];
        }
    }
}

extern "C" hipError_t isk_launch_cluster(int n_slots, float eps, int min_pts, int n_images,
                                         const is_instance_buffers* d_tbl,
                                         const is_instance_buffers* one, int32_t* scratch,
                                         hipStream_t stream) {
    is_instance_buffers o;
    if (one) o = *one; else o = is_instance_buffers{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(k_cluster_instances, dim3(IS_INSTANCE_CLASSES, n_images), dim3(CLU_THREADS), 0,
                       stream, n_slots, eps * eps, min_pts, d_tbl, o, scratch);
    return hipGetLastError();
}
