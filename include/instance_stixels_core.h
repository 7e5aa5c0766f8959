/*
 * instance_stixels_core.h -- C ABI of the MI355X (gfx950) Instance-Stixels column-DP core.
 *
 * This is the drop-in boundary of the hot path: plain pointers and sizes, no C++ or torch
 * types.  It is what the reference's host class `Stixels`
 * (/root/reference/InstanceStixels/include/InstanceStixels/Stixels.hpp:40-96) needs from the
 * device side: the three kernel launches of `Stixels::Compute`
 * (/root/reference/InstanceStixels/src/Stixels.cu:509-590) plus the buffer management of
 * `Stixels::Initialize` / `Finish` (Stixels.cu:43-283).  The header-compatible `Stixels`
 * class in include/InstanceStixels/Stixels.hpp is implemented only in terms of these
 * entry points.
 *
 * All functions return 0 on success and a negative IS_E* code on failure; the failing HIP
 * error string is available from is_last_error().  Nothing here falls back to a CPU path.
 */
#ifndef INSTANCE_STIXELS_CORE_H_
#define INSTANCE_STIXELS_CORE_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IS_OK 0
#define IS_EINVAL (-1)  /* bad argument / unsupported shape */
#define IS_EHIP (-2)    /* HIP runtime failure, see is_last_error() */
#define IS_ENOMEM (-3)

#define IS_GROUND 0
#define IS_OBJECT 1
#define IS_SKY 2

#define IS_DOWNSAMPLE_FACTOR 8       /* configuration.h:31 */
#define IS_MAX_STIXELS_PER_COLUMN 200 /* configuration.h:32 */
#define IS_INSTANCE_CLASSES 8        /* Stixels.cu:47 (Cityscapes classes 11..18) */
#define IS_FIRST_INSTANCE_CLASS 11   /* StixelsKernels.cu:926 */

/* Same fields, order and layout as `struct StixelParameters`
 * (/root/reference/InstanceStixels/include/InstanceStixels/types.h:145-184). */
typedef struct is_stixel_params {
    int vhor;
    int rows;
    int rows_power2;
    int rows_power2_segmentation;
    int cols; /* = realcols, Stixels.cu:213 */
    int max_dis;
    float rows_log;
    float pnexists_given_sky_log;
    float normalization_sky;
    float inv_sigma2_sky;
    float puniform_sky;
    float nopnexists_given_sky_log;
    float pnexists_given_ground_log;
    float puniform;
    float nopnexists_given_ground_log;
    float pnexists_given_object_log;
    float nopnexists_given_object_log;
    float baseline;
    float focal;
    float range_objects_z;
    float pord;
    float epsilon;
    float pgrav;
    float pblg;
    float max_dis_log;
    int max_sections;
    int width_margin;
    int segmentation_classes;
    int segmentation_channels;
    float prior_weight;
    float disparity_weight;
    float segmentation_weight;
    float instance_weight;
    int column_step;
    float clustering_eps;
    int clustering_min_pts;
    int clustering_size_filter;
    float invalid_disparity;
} is_stixel_params;

/* Same layout as `struct Section` (types.h:186-194), 32 bytes. */
typedef struct is_section {
    int type; /* IS_GROUND / IS_OBJECT / IS_SKY, -1 = terminator */
    int vB, vT;
    float disparity;
    int semantic_class;
    float cost;
    float instance_meanx;
    float instance_meany;
} is_section;

/* Device-side instance-candidate arrays of ONE image, laid out as the reference's
 * d_instance_* members (Stixels.cu:56-74; filled at StixelsKernels.cu:926-942).
 * class_offset = class_id * realcols * max_sections.  Any pointer may be NULL to skip it.
 * Order inside a class is canonical (column ascending, then section index ascending)
 * instead of the reference's atomic arrival order. */
/* ABI 0.3: the struct MUST be zero-initialised (memset / `= {}`) before its fields are set --
 * is_compute dereferences every non-NULL member, and members added by later versions then stay
 * NULL.  All calls on one context must be stream-ordered (the context owns scratch that
 * is_compute and is_cluster_instances share). */
typedef struct is_instance_buffers {
    float* d_centerofmass;     /* [8][realcols*max_sections][2]  (meanx, meany) */
    int32_t* d_indices;        /* [8][realcols*max_sections][2]  (column, section index) */
    uint8_t* d_core_candidates; /* [8][realcols*max_sections]    (bool)   */
    int32_t* d_instances_per_class; /* [8] */
    /* [8][realcols*max_sections] cluster label of every candidate (the reference's
     * d_instance_labels, Stixels.cu:660-666): 0.. per class, -1 = no instance.  NULL skips the
     * clustering; non-NULL requires the three arrays above. */
    int32_t* d_labels;
    /* optional, [1 + 3*8*realcols*max_sections]: d_packed[0] = number of candidates of all
     * classes, then one (column, section index, label) triple per candidate, classes ascending:
     * what Stixels::GetInstanceStixels (Stixels.cu:744-776) needs, in one small copy.  Written
     * together with d_labels; needs d_indices. */
    int32_t* d_packed;
} is_instance_buffers;

typedef struct is_ctx is_ctx;

/* Replaces the device half of Stixels::Initialize (Stixels.cu:53-74, 136-210): uploads the
 * frame-independent LUTs and allocates all scratch for up to `max_batch` images per call.
 *   obj_cost_lut         host, [max_dis][max_dis]   (Stixels.cu:122-129)
 *   obj_disparity_range  host, [max_dis]            (Stixels.cu:111-115)
 * `params->vhor` is ignored here (it is a per-frame value, Stixels.cu:532). */
int is_ctx_create(const is_stixel_params* params, const float* obj_cost_lut,
                  const float* obj_disparity_range, int max_batch, int device,
                  is_ctx** out_ctx);

/* Replaces Stixels::Finish (Stixels.cu:250-283). */
int is_ctx_destroy(is_ctx* ctx);

/* Replaces the JoinColumns launch (Stixels.cu:509-511, kernel StixelsKernels.cu:980-1095).
 *   d_disparity_big  device, [n_images][rows][full_cols] row-major, image row 0 = top
 *   d_joined         device, [n_images][realcols][rows], row 0 = image bottom
 * `stream` is a hipStream_t (NULL = default stream). */
int is_join_columns(is_ctx* ctx, const float* d_disparity_big, int full_cols, int median_join,
                    float* d_joined, int n_images, void* stream);

/* Replaces the ComputeObjectLUT + StixelsKernel<PAIRWISE> launches (Stixels.cu:535-590) for
 * a batch of `n_images` independent images that share the configuration of the context.
 *   d_joined            device, [n_images][realcols][rows]                 (read only)
 *   d_segmentation      device, [n_images][realcols][channels][rows_power2_segmentation]
 *                       int32, layout of tools/CNN_training/models/wrappers.py:35-61.  Read
 *                       only: unlike the reference the input is NOT modified.
 *   h_ground_function, h_normalization_ground, h_inv_sigma2_ground
 *                       host, [n_images][rows]  (Stixels::PrecomputeGround, Stixels.cu:790-817)
 *   h_vhor              host, [n_images], library convention rows-vhor_image-1 (Stixels.cu:377)
 *   d_sections          device, [n_images][realcols][max_sections] is_section
 *   instances           per image (array of n_images) or NULL.  The candidates of the whole
 *                       batch are compacted by ONE launch and clustered by ONE launch
 *                       (grid = 8 classes x n_images), whatever n_images is.
 *   d_cost_table        optional device out, [n_images][realcols][rows][3] final DP costs
 *   d_index_table       optional device out, [n_images][realcols][rows][3] int32.  PAIRWISE:
 *                       vB*3 + predecessor type, the reference's encoding
 *                       (StixelsKernels.cu:723-727).  UNARY: the winning vB only (-1 = no
 *                       candidate) -- in unary mode the predecessor type depends only on the
 *                       final cost_table[vB-1] and is resolved by the back-trace,
 *                       so decoding a unary table with /3 and %3 is WRONG.
 * Alignment: d_joined and d_segmentation are read with 16-byte vector loads; both must be
 * 16-byte aligned (any hipMalloc / torch allocation is) -- checked, IS_EINVAL otherwise.
 * Device: the call runs on the context's device whatever the caller's current device is (the
 * current device is restored on return); `stream` must belong to that device.
 * Asynchronous with respect to the host: work is queued on `stream`; the host arrays are copied
 * into a ring of pinned staging slots before the call returns.
 * Internal invariant: the context's generic-column counter is zero between calls (counted by the
 * prepare kernel, cleared by the back-trace at the end of the call); a call that returns an error
 * clears it itself. */
int is_compute(is_ctx* ctx, const float* d_joined, const int32_t* d_segmentation,
               const float* h_ground_function, const float* h_normalization_ground,
               const float* h_inv_sigma2_ground, const int* h_vhor, int pairwise,
               int n_images, is_section* d_sections, const is_instance_buffers* instances,
               float* d_cost_table, int32_t* d_index_table, void* stream);

/* Replaces Stixels::ClusterInstances (Stixels.cu:639-681: one ML::dbscanFit per instance class
 * with the size filter of the cuML fork) for the candidates of ONE image (is_compute clusters a
 * whole batch in one launch): size-filtered DBSCAN
 * with the semantics of the reference's Python twin
 * (tools/visualization/clustering_visualization.py:894-960) on the device, no host round trip.
 * Reads d_centerofmass / d_core_candidates / d_instances_per_class, writes d_labels.
 * is_compute() runs it by itself for every image whose d_labels is set. */
int is_cluster_instances(is_ctx* ctx, const is_instance_buffers* instances, void* stream);

/* Compaction of the fixed-stride Section output for the final gather of a multi-GPU batch
 * (the reference copies all max_sections = 200 slots of every column to the host,
 * Stixels.cu:629-633, of which 10-40 are used).  On the current device, on `stream`:
 *   d_sections  [n_columns][max_sections] (n_columns = n_images * realcols), terminator type -1
 *   d_counts    [n_columns]      sections in front of each column's terminator
 *   d_offsets   [n_columns + 1]  exclusive prefix of the counts; d_offsets[n_columns] = total
 *   d_packed    [>= total]       the sections in (image, column, section) order
 * is_unpack_sections is the inverse (it recomputes d_offsets from d_counts and writes the
 * terminators); entries behind a terminator are unspecified on both sides. */
int is_pack_sections(const is_section* d_sections, int n_columns, int max_sections,
                     int32_t* d_counts, int32_t* d_offsets, is_section* d_packed, void* stream);
int is_unpack_sections(const int32_t* d_counts, int32_t* d_offsets, const is_section* d_packed,
                       int n_columns, int max_sections, is_section* d_sections, void* stream);

/* ---- multi-GPU: the final gather of a sharded batch for C / C++ callers ----------------
 * Images are independent: every rank (one process per GPU) runs is_compute on its shard of the batch and
 * only the OUTPUT travels, to one rank, over RCCL (/opt/rocm/include/rccl/rccl.h:700-745).  `comm` is an
 * ncclComm_t passed as void*; a caller that has rccl.h uses its own communicator, a plain-C++ caller the
 * four helpers below.  RCCL is loaded at run time (dlopen of librccl.so.1): IS_EHIP if it is missing.
 * There is no reference counterpart: the reference runs on one GPU and writes its Sections to a file
 * per frame (apps/run_cityscapes.cu:430-449). */
int is_comm_unique_id(void* id_out, size_t id_bytes);  /* ncclGetUniqueId; id_bytes >= 128; pass the bytes to every rank */
int is_comm_init_rank(void** comm, int nranks, const void* id, int rank); /* ncclCommInitRank on the current device */
int is_comm_destroy(void* comm);
int is_comm_rank(void* comm, int* rank, int* nranks);
/* Gather of int32 payloads of different sizes on rank `dst`: rank r sends h_counts[r] elements of d_send;
 * dst receives them into d_recv in rank order (its own part is a device copy).  h_counts has one entry per
 * rank; a sender reads only its own.  Grouped ncclSend / ncclRecv on `stream`, asynchronous. */
int is_gather_i32(void* comm, int dst, const int64_t* h_counts, const int32_t* d_send, int32_t* d_recv,
                  void* stream);
/* The compacted gather: what is_pack_sections left on every rank goes to `dst`.
 *   h_columns      host, [nranks]: columns (n_images * realcols) of every rank's shard -- the shard sizes are
 *                  known everywhere (contiguous blocks of the batch); the same array on every rank
 *   d_counts, d_offsets, d_packed   this rank's is_pack_sections output (d_offsets[columns] = its total)
 *   d_all_counts   dst: [sum of h_columns], rank order
 *   d_all_packed   dst: [cap_sections] is_section, the ranks' sections back to back in rank order
 *   h_totals       host, [nranks], out: on dst the sections of every rank, elsewhere only entry [own rank]
 * Two phases on `stream`: the sizes (ncclGather) and the per-column counts, a host synchronisation, dst's
 * go-ahead (IS_ENOMEM on EVERY rank when the total exceeds cap_sections: nothing is sent), then the payload
 * (queued; the caller synchronises `stream`).  is_unpack_sections on dst restores the fixed-stride arrays. */
int is_gather_sections(void* comm, int dst, const int32_t* h_columns, const int32_t* d_counts,
                       const int32_t* d_offsets, const is_section* d_packed, int32_t* d_all_counts,
                       is_section* d_all_packed, size_t cap_sections, int64_t* h_totals, void* stream);

/* Replaces the output wrapper of the reference's CNN export ("FlipAndPad",
 * tools/CNN_training/models/wrappers.py:35-61), i.e. the producer of d_segmentation:
 *   d_cnn_out       device, [n_images][channels][rows8][cols8] float (NCHW network output)
 *   d_segmentation  device, [n_images][cols8][channels][rows_power2_segmentation] int32:
 *                   permuted, rows flipped (index 0 = image bottom), zero padded, (int)(8*x). */
int is_flip_and_pad(const float* d_cnn_out, int32_t* d_segmentation, int n_images, int channels,
                    int rows8, int cols8, int rows_power2_segmentation, void* stream);

/* Replaces the three launches of RoadEstimation::Compute (RoadEstimation.cu:103-118, kernels
 * RoadEstimationKernels.cu:25-60): v-disparity histogram of the full-resolution disparity image
 * (pixels equal to 0 are skipped), its maximum, and the binarised image
 * (count > maximum * threshold ? 255 : 0).
 *   d_disparity [rows][cols] float;  d_vdisp [rows][max_dis] int;  d_maximum [1] int;
 *   d_binary [rows][max_dis] uint8. */
int is_road_vdisparity(const float* d_disparity, int rows, int cols, int max_dis, float threshold,
                       int* d_vdisp, int* d_maximum, uint8_t* d_binary, void* stream);

/* Thin wrappers over the HIP runtime so that the plain-C++ host class needs no HIP headers
 * (the reference's callers are all .cu files; ours may be plain C++). */
int is_device_malloc(void** ptr, size_t bytes);
int is_device_free(void* ptr);
int is_host_malloc(void** ptr, size_t bytes); /* pinned host memory */
int is_host_free(void* ptr);
int is_get_device(int* device);               /* the calling thread's current HIP device */
int is_set_device(int device);
int is_ctx_device(const is_ctx* ctx);         /* the device a context lives on */
int is_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream);
int is_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream);
/* `height` rows of `width` bytes from a pitched device array into a pitched (pinned) host array:
 * the host class fetches the first sections of every column this way instead of all max_sections */
int is_memcpy2d_d2h(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height,
                    void* stream);
int is_memset(void* dst, int value, size_t bytes, void* stream);
int is_stream_synchronize(void* stream);
/* A stream of the current device.  blocking != 0: an ordinary stream that still synchronises
 * implicitly with the legacy NULL stream, like every stream the reference's callers create
 * (cudaStreamCreate); 0: hipStreamNonBlocking. */
int is_stream_create(void** stream, int blocking);
int is_stream_destroy(void* stream);
int is_device_synchronize(void);

/* Test hook (A4, ComputeObjectLUT): copies the object data-cost prefix table of ONE stixel column
 * (0 <= column < n_images * realcols) as the last is_compute call on this context left it into
 * host memory, h_out[(rows + 1) * max_dis] = lutT[v][fn] -- the transpose of the reference's
 * d_object_lut[fn][v] (Stixels.cu:159-160, StixelsKernels.cu:959-978).  Synchronises the device.  (A context
 * created with IS_LUT_CARRY=1 materialises only the rows 32 k of unary calls whose every tile is windowed: the
 * other rows then hold what an earlier call left there.) */
int is_debug_read_object_lut(is_ctx* ctx, int column, float* h_out);

/* Test hook: did the last unary is_compute call on this context run its repair launches?  A unary call whose every
 * tile is windowed builds the object data-cost table INSIDE its DP launch (the units of a column ahead of the
 * column's DP workgroups, which wait for a per-column count); a DP workgroup that cannot trust the hand-over -- its
 * column's units ran on another XCD than itself, or did not finish within the bound of its poll -- sets a word, and
 * the two launches queued behind (the ordinary table kernel and the ordinary DP launch, which otherwise leave at once)
 * do the call again.  *repaired = that word (0 in normal operation; IS_LUT_FUSED=2 forces 1 for tests, IS_LUT_FUSED=0
 * keeps the table in the prepare launch).  Synchronises the device. */
int is_debug_lut_fused_state(is_ctx* ctx, int* repaired);

/* The number of is_compute calls of this context whose repair launches have run since it was created (the reference
 * gets the order of its two launches from the stream, Stixels.cu:535-590; the fused launch has to earn it).  After the
 * first one the context plans its later calls with the table in the prepare launch again (unless IS_LUT_FUSED is set
 * to 1 or 2).  Synchronises the device. */
int is_lut_fused_repairs(is_ctx* ctx, int* calls_repaired);

/* Test hook: the bound-block summaries the pairwise DP of the last is_compute call left for one stixel
 * column (lemmas L7 / L8): h_out[n_blocks][24], returns n_blocks through *n_blocks. */
int is_debug_read_block_summaries(is_ctx* ctx, int column, float* h_out, int cap_floats, int* n_blocks);

/* Introspection used by bench.py / tests. */
const char* is_last_error(void);
const char* is_version(void);
/* Average duration (ms) of the DP kernel launches recorded by the last is_compute call on
 * this context, measured with HIP events on the launch stream; <0 if timing is disabled. */
int is_set_kernel_timing(is_ctx* ctx, int enabled);
int is_get_kernel_times_ms(is_ctx* ctx, float* prepare_ms, float* dp_ms, float* backtrace_ms);
size_t is_scratch_bytes(const is_ctx* ctx);
/* Evaluation counters of the exact branch-and-bound, for measurements
 * OUTSIDE a timed region: while enabled, the DP kernels of FAST columns add the number of 64-pair
 * wave-steps they evaluated below the diagonal blocks to a device array (enabling resets it).
 *   out[0] unary full steps, out[1] unary ground/sky-only steps,
 *   out[2] pairwise phase-1 full steps, out[3] pairwise phase-1 ground/sky-only candidates,
 *   out[4] pairwise phase-1 steps in which some lane read OUTSIDE its fn window (window misses),
 *   out[5] the same for the unary ring kernel; out[6] / out[7]: the fused LUT units of the unary launch (polls of
 *   waiting DP workgroups / shader clocks the units lived);
 *   out[8 + 3 t + j], t < 64: the phase-1 launch of 64-row tile t alone, j = 0 full, 1 window
 *   misses, 2 ground/sky-only.  n <= IS_EVAL_COUNTERS.  Both calls synchronise the device. */
#define IS_EVAL_COUNTERS 200
int is_set_eval_counters(is_ctx* ctx, int enabled);
int is_get_eval_counters(is_ctx* ctx, unsigned long long* out, int n);

#ifdef __cplusplus
}
#endif

#endif /* INSTANCE_STIXELS_CORE_H_ */
