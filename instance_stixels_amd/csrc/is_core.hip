/*
 * is_core.hip -- C ABI of the gfx950 column-DP core (include/instance_stixels_core.h).
 *
 * Owns what the device half of the reference's Stixels::Initialize / Compute / Finish owns
 * (/root/reference/InstanceStixels/src/Stixels.cu:43-283, 449-637): frame-independent LUTs,
 * per-column scratch (boundary records + object LUT), DP tables, and the launch sequence.
 * There is no CPU fallback: every failure is reported through the return code.
 */
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "instance_stixels_core.h"
#include "is_device.h"
#include "is_numerics.h"

extern "C" {
size_t isk_prepare_lds_bytes(const DevParams* P);
size_t isk_unary_lds_bytes(const DevParams* P);
size_t isk_pairwise_lds_bytes(const DevParams* P, int nwaves);
hipError_t isk_launch_join(const float*, float*, int, int, int, int, int, int, float, int, hipStream_t);
hipError_t isk_launch_prepare(const DevParams*, int, const float*, const int32_t*, const float*,
                              const int*, const float*, RowRec*, float*, int*, float*, PruneRec*, int*,
                              hipStream_t, hipStream_t, hipEvent_t, hipEvent_t);
struct StepRec;
hipError_t isk_launch_priors(const DevParams*, const float*, PriorRec*, int, hipStream_t);
hipError_t isk_launch_dp_unary(const DevParams*, int, int, const RowRec*, const float*, const float*,
                               const int*, const int*, const PruneRec*, float*, int32_t*, const int*,
                               unsigned long long*, const float*, const float*, hipStream_t);
hipError_t isk_launch_dp_pairwise(const DevParams*, int, int, const RowRec*, const float*,
                                  const float*, const PriorRec*, const float*, const float*, const float*,
                                  const int*, const int*, const PruneRec*, StepRec*, float*, int*,
                                  float*, int32_t*, unsigned long long*, const float*, const int*, float*, float*,
                                  hipStream_t, hipStream_t*, int, hipEvent_t, hipEvent_t*);
hipError_t isk_launch_backtrace(const DevParams*, int, int, const RowRec*, const float*,
                                const int32_t*, const int*, is_section*, int*, int*, hipStream_t);
hipError_t isk_launch_compact(const DevParams*, int, const is_section*, const int*,
                              const is_instance_buffers*, hipStream_t);
hipError_t isk_set_lds_prepare(const DevParams*);
hipError_t isk_set_lds_unary(const DevParams*);
hipError_t isk_set_lds_pairwise(const DevParams*, int);
hipError_t isk_set_lds_backtrace(const DevParams*);
int isk_debug_occupancy(const DevParams*, int);
int isk_unary_uses_carry(const DevParams*, int);
int isk_unary_uses_fused_lut(const DevParams*, int);
hipError_t isk_launch_cluster(int, float, int, int, const is_instance_buffers*,
                              const is_instance_buffers*, int32_t*, hipStream_t);
size_t isk_phase2_lds_bytes(const DevParams* P);
size_t isk_phase2s_lds_bytes(const DevParams* P);
hipError_t isk_launch_pack(const is_section*, int, int, int32_t*, int32_t*, is_section*, hipStream_t);
hipError_t isk_launch_unpack(const int32_t*, int32_t*, const is_section*, int, int, is_section*, hipStream_t);
hipError_t isk_launch_flip_and_pad(const float*, int32_t*, int, int, int, int, int, hipStream_t);
hipError_t isk_launch_vdisparity(const float*, int*, int*, uint8_t*, int, int, int, float, hipStream_t);
}

#define IS_FLT_HUGE 1e30f
static thread_local char g_err[512] = "";

static int fail_hip(hipError_t e, const char* what, const char* file, int line) {
    snprintf(g_err, sizeof(g_err), "%s returned %s (%d) at %s:%d", what, hipGetErrorString(e),
             (int)e, file, line);
    return IS_EHIP;
}
static int fail_arg(const char* msg) {
    snprintf(g_err, sizeof(g_err), "invalid argument: %s", msg);
    return IS_EINVAL;
}
/* (for the other translation units of the library: is_gather.hip) */
extern "C" int isk_fail(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}
#define HIP_TRY(expr)                                                    \
    do {                                                                 \
        hipError_t e__ = (expr);                                         \
        if (e__ != hipSuccess) return fail_hip(e__, #expr, __FILE__, __LINE__); \
    } while (0)

#define IS_STAGE_SLOTS 4 /* pinned staging ring of the per-frame ground model */
#define IS_GRAPH_MAX_IMAGES 8 /* calls of up to that many images are replayed as hipGraphs */
#define IS_GRAPH_ENTRIES 4    /* distinct argument sets remembered per context */
struct is_graph_entry {
    bool valid;
    const void *joined, *seg, *sections, *ct, *it;
    int pairwise, n_images, n_inst;
    int win_tiles; /* the windowed / classic tile split of the captured launches (a function of the call's horizons) */
    is_instance_buffers inst[IS_GRAPH_MAX_IMAGES];
    int slot; /* the pinned staging slot baked into the copy nodes */
    hipGraphExec_t exec;
    unsigned long long last_use;
};


struct is_ctx {
    is_stixel_params params;
    DevParams dp;
    int device;
    int max_batch;
    int nwaves_unary, nwaves_pairwise;
    /* frame-independent device tables */
    float* d_obj_cost_lut;   /* [D dis][D fn], transposed w.r.t. Stixels.cu:122-129 */
    float* d_odr;            /* [D]     object_disparity_range */
    float* d_rcp;            /* [H+1]   RN(1/h) = (float)(1./h), the reference's inverse_height */
    int* d_col_flags;        /* [max_batch*C] 0 = FAST column, see RowRec */
    PruneRec* d_prune;       /* [max_batch*C] branch-and-bound slacks of the column */
    int* d_n_generic;        /* [1] generic-encoding columns of the current call */
    /* per-call device inputs */
    /* one block [ground: max_batch x 3 x H floats][instance table: max_batch][vhor: max_batch ints], on
     * the device and in every pinned staging slot: a full batch (the host class's single frame
     * included) travels in ONE copy */
    char* d_stage;
    size_t stage_bytes, stage_off_inst, stage_off_vhor;
    char* h_stage[IS_STAGE_SLOTS];
    float* d_ground;         /* [max_batch][3][H]            (in d_stage) */
    int* d_vhor;             /* [max_batch]                  (in d_stage) */
    /* ring of pinned staging slots: a call blocks the host only when the slot it wants is still
     * being read by the H2D copy of the call IS_STAGE_SLOTS calls ago */
    float* h_ground_pinned[IS_STAGE_SLOTS];
    int* h_vhor_pinned[IS_STAGE_SLOTS];
    is_instance_buffers* h_inst_pinned[IS_STAGE_SLOTS]; /* [max_batch] per-image output arrays */
    hipEvent_t staging_free[IS_STAGE_SLOTS]; /* recorded after the H2D copies of the slot's call */
    bool staging_pending[IS_STAGE_SLOTS];
    int stage_next;
    hipStream_t aux_stream;  /* = aux_streams[0]: second stream of the prepare kernels */
    hipStream_t aux_streams[IS_AUX_STREAMS]; /* column groups of the pairwise DP in flight */
    hipEvent_t ev_fork, ev_join;
    hipEvent_t ev_joins[IS_AUX_STREAMS];
    int32_t* d_cluster_scratch; /* [max_batch][8][2][C*S] work arrays of k_cluster_instances */
    is_instance_buffers* d_inst_tbl; /* [max_batch] device copy of the caller's per-image arrays */
    int* d_inst_cnt;            /* [max_batch*C][8] instance candidates per column and class */
    unsigned long long* d_counters; /* [IS_CNT_N] evaluation counters (is_set_eval_counters) */
    bool counting;
    /* hipGraph replay of small calls (see is_compute) */
    bool graphs;
    unsigned long long graph_clock;
    struct is_graph_entry* graph_cache; /* [IS_GRAPH_ENTRIES] */
    /* scratch */
    RowRec* d_recs;          /* [max_batch*C][H+1] */
    float* d_lutT;           /* [max_batch*C][H+1][D] */
    int* h_lutf_repairs = nullptr; /* pinned + mapped: calls whose fused LUT hand-over was repaired (DevParams::lutf_repairs) */
    PriorRec* d_priors;      /* [max_batch][H] */
    StepRec* d_steps;        /* [max_batch*C][H]   per-vB transition records of the pairwise DP (64 B) */
    float* d_part_cost;      /* [max_batch*C][3][64] merged partial minima of the current tile */
    int* d_part_idx;         /* [max_batch*C][3][64] */
    float* d_sv;             /* [max_batch*C][2][H+1] compact S / V prefixes */
    float* d_t8row;          /* [max_batch*C][H] pw * (smallest p field) of every StepRec (lemma L8) */
    float* d_blksum;         /* [max_batch*C][ntiles*IS_QPT+1][24] bound-block summaries of the pairwise DP (lemmas L7, L8) */
    float* d_cost_table;     /* [max_batch*C][H][3] */
    int32_t* d_index_table;  /* [max_batch*C][H][3] */
    size_t scratch_bytes;
    /* timing */
    bool timing;
    hipEvent_t ev[4];
    bool ev_valid;
};

static int ilog2_exact(int n) {
    int l = 0;
    while ((1 << l) < n) l++;
    return l;
}

/* Runs the enclosed calls on the context's device and puts the caller's current device back. */
struct DeviceScope {
    int prev = -1, want;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int device) : want(device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != want) {
            err = hipSetDevice(want);
            switched = err == hipSuccess;
        }
    }
    ~DeviceScope() {
        if (switched) (void)hipSetDevice(prev);
    }
};
#define ON_CTX_DEVICE(c)                                                         \
    DeviceScope dev_scope__((c)->device);                                        \
    if (dev_scope__.err != hipSuccess)                                           \
        return fail_hip(dev_scope__.err, "hipSetDevice(ctx->device)", __FILE__, __LINE__)

const char* is_last_error(void) { return g_err; }
const char* is_version(void) { return "instance_stixels_amd-core 0.3 (gfx950)"; }

int is_device_malloc(void** ptr, size_t bytes) { HIP_TRY(hipMalloc(ptr, bytes)); return IS_OK; }
int is_device_free(void* ptr) { HIP_TRY(hipFree(ptr)); return IS_OK; }
int is_host_malloc(void** ptr, size_t bytes) { HIP_TRY(hipHostMalloc(ptr, bytes)); return IS_OK; }
int is_host_free(void* ptr) { HIP_TRY(hipHostFree(ptr)); return IS_OK; }
int is_get_device(int* device) {
    if (!device) return fail_arg("null pointer");
    HIP_TRY(hipGetDevice(device));
    return IS_OK;
}
int is_set_device(int device) { HIP_TRY(hipSetDevice(device)); return IS_OK; }
int is_ctx_device(const is_ctx* ctx) { return ctx ? ctx->device : -1; }
int is_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return IS_OK;
}
int is_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return IS_OK;
}
int is_memcpy2d_d2h(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height,
                    void* stream) {
    HIP_TRY(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return IS_OK;
}
int is_memset(void* dst, int value, size_t bytes, void* stream) {
    HIP_TRY(hipMemsetAsync(dst, value, bytes, (hipStream_t)stream));
    return IS_OK;
}
int is_stream_synchronize(void* stream) { HIP_TRY(hipStreamSynchronize((hipStream_t)stream)); return IS_OK; }
int is_device_synchronize(void) { HIP_TRY(hipDeviceSynchronize()); return IS_OK; }
int is_stream_create(void** stream, int blocking) {
    if (!stream) return fail_arg("null pointer");
    hipStream_t s;
    HIP_TRY(hipStreamCreateWithFlags(&s, blocking ? hipStreamDefault : hipStreamNonBlocking));
    *stream = (void*)s;
    return IS_OK;
}
int is_stream_destroy(void* stream) {
    if (stream) HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return IS_OK;
}

size_t is_scratch_bytes(const is_ctx* ctx) { return ctx ? ctx->scratch_bytes : 0; }

static int ctx_init(is_ctx* c, const is_stixel_params* p, const float* obj_cost_lut,
                    const float* obj_disparity_range, int max_batch, int device, int P2, int P2S);

int is_ctx_create(const is_stixel_params* p, const float* obj_cost_lut,
                  const float* obj_disparity_range, int max_batch, int device, is_ctx** out_ctx) {
    if (!p || !obj_cost_lut || !obj_disparity_range || !out_ctx) return fail_arg("null pointer");
    if (max_batch < 1) return fail_arg("max_batch < 1");
    if (p->column_step != IS_DOWNSAMPLE_FACTOR)
        return fail_arg("column_step must be 8 (assert at StixelsKernels.cu:318)");
    if (p->rows < 8 || p->rows % 8 != 0) return fail_arg("rows must be a positive multiple of 8");
    if (p->rows * 3 + 2 >= 32768) return fail_arg("rows too large for the int16 index convention");
    if (p->max_dis < 2 || p->max_dis > 1024) return fail_arg("max_dis out of range [2, 1024]");
    if (p->segmentation_classes != 19 || p->segmentation_channels != 21)
        return fail_arg("the class model is Cityscapes: 19 classes + 2 offset channels (Cityscapes.h)");
    if (p->cols < 1) return fail_arg("cols (realcols) < 1");
    if (p->max_sections < 2) return fail_arg("max_sections < 2");
    const int P2 = 1 << ilog2_exact(p->rows + 1);
    const int P2S = 1 << ilog2_exact(p->rows / 8 + 1);
    if (p->rows_power2 != P2 || p->rows_power2_segmentation != P2S)
        return fail_arg("rows_power2 / rows_power2_segmentation inconsistent with rows (Stixels.cu:131-133)");

    /* everything of the context is created on `device`; the caller's current device is put back */
    DeviceScope scope(device);
    if (scope.err != hipSuccess) return fail_hip(scope.err, "hipSetDevice(device)", __FILE__, __LINE__);
    is_ctx* c = (is_ctx*)calloc(1, sizeof(is_ctx));
    if (!c) return IS_ENOMEM;
    const int rc = ctx_init(c, p, obj_cost_lut, obj_disparity_range, max_batch, device, P2, P2S);
    if (rc != IS_OK) { /* release whatever was created; keep the error text of the failure */
        char keep[sizeof(g_err)];
        memcpy(keep, g_err, sizeof(keep));
        is_ctx_destroy(c);
        memcpy(g_err, keep, sizeof(keep));
        return rc;
    }
    *out_ctx = c;
    return IS_OK;
}

static int ctx_init(is_ctx* c, const is_stixel_params* p, const float* obj_cost_lut,
                    const float* obj_disparity_range, int max_batch, int device, int P2, int P2S) {
    c->params = *p;
    c->device = device;
    c->max_batch = max_batch;
    c->graphs = getenv("IS_GRAPH") != nullptr; /* opt-in: measured slower than eager launches (see is_compute) */
    c->graph_cache = (is_graph_entry*)calloc(IS_GRAPH_ENTRIES, sizeof(is_graph_entry));
    if (!c->graph_cache) return IS_ENOMEM;

    DevParams& d = c->dp;
    d.H = p->rows; d.C = p->cols; d.D = p->max_dis; d.P2 = P2; d.P2S = P2S;
    d.CH = p->segmentation_channels; d.K = p->segmentation_classes; d.S = p->max_sections;
    d.ntiles = (d.H + IS_TILE - 1) / IS_TILE;
    d.log2P2 = ilog2_exact(P2);
    d.invalid = p->invalid_disparity;
    d.pnex_sky_log = p->pnexists_given_sky_log; d.norm_sky = p->normalization_sky;
    d.inv_sigma2_sky = p->inv_sigma2_sky; d.puniform_sky = p->puniform_sky;
    d.nopnex_sky_log = p->nopnexists_given_sky_log;
    d.pnex_gnd_log = p->pnexists_given_ground_log; d.puniform = p->puniform;
    d.nopnex_gnd_log = p->nopnexists_given_ground_log;
    d.dw = p->disparity_weight; d.pw = p->prior_weight; d.sw = p->segmentation_weight;
    d.iw = p->instance_weight;
    d.rows_log = p->rows_log; d.max_dis_log = p->max_dis_log; d.epsilon = p->epsilon;
    d.pgrav = p->pgrav; d.pblg = p->pblg; d.pord = p->pord; d.max_disf = (float)p->max_dis;
    d.log2c = is_logf(2.0f);
    d.nlog07 = -is_logf(0.7f);
    d.nlog03 = -is_logf(0.3f);
    d.nlog_pord = -is_logf(p->pord);
    d.nlog_1mpord = -is_logf(1.0f - p->pord);
    d.first_g = d.log2c + d.rows_log;                       /* StixelsKernels.cu:196-199 */
    d.first_o_below = d.rows_log + d.log2c + d.max_dis_log; /* :189-194 */
    d.first_o_above = d.rows_log + 0.0f + d.max_dis_log;
    d.size_filter = p->clustering_size_filter;
    d.column_step = p->column_step;
    /* the IS_* knobs (experiments, A/B tests) are read here, once per context, never per call */
    const bool no_prune = getenv("IS_NO_PRUNE") != nullptr;
    const bool debug = getenv("IS_DEBUG") != nullptr;
    {
        auto knob = [](const char* name) { const char* e = getenv(name); return e ? atoi(e) : -1; };
        d.knob_ring_kernel = getenv("IS_NO_RING_KERNEL") ? 0 : -1;
        d.knob_prepare_overlap = knob("IS_PREPARE_OVERLAP");
        d.knob_p2_lds_floor = knob("IS_P2_LDS");
        d.knob_pw_groups = knob("IS_PW_GROUPS");
        d.knob_p2_split = knob("IS_P2_SPLIT");
        d.knob_p2x = knob("IS_P2X");
        d.knob_unary_diag = knob("IS_UNARY_DIAG") == 1; /* (experiment: off unless asked for) */
        d.knob_win_tiles = knob("IS_P1_WIN_TILES");
        d.knob_pw_waves = knob("IS_PW_WAVES");
        d.knob_lut_carry = knob("IS_LUT_CARRY"); /* carry-only lutT (is_device.h): opt-in */
        d.knob_lut_fused = knob("IS_LUT_FUSED"); /* the LUT units inside the unary DP launch (is_k_unary_fast.hip, LUTF) */
    }
    {
        /* branch-and-bound constants (PruneRec, is_device.h).  gamma_d bounds the relative error of
         * a prefix computed by a summation tree of depth d: Blelloch needs <= 2 log2(P2) additions
         * on a path, the object LUT's carry chain H/32 + 5; a generous d covers both. */
        const double depth = 2.0 * ilog2_exact(P2) + (double)d.H / 32.0 + 8.0;
        const double gamma = 1.01 * depth * 0x1p-24;
        d.gamma2 = (float)(2.0 * gamma * 1.001);
        double max_abs = 0.0, min_v = 0.0;
        bool finite = true;
        for (size_t i = 0; i < (size_t)d.D * d.D; i++) {
            const double v = obj_cost_lut[i];
            if (!(fabs(v) < 1e30)) finite = false; /* also catches NaN */
            if (fabs(v) > max_abs) max_abs = fabs(v);
            if (v < min_v) min_v = v;
        }
        const bool weights_ok = d.dw >= 0.0f && d.sw >= 0.0f && d.iw >= 0.0f && d.pw >= 0.0f &&
                                d.dw < IS_FLT_HUGE && d.sw < IS_FLT_HUGE && d.iw < IS_FLT_HUGE;
        if (finite && weights_ok && !no_prune)
            d.sigma_od = (float)(((0.0 - min_v) * d.H + 2.0 * gamma * d.H * max_abs) * 1.001);
        else
            d.sigma_od = __builtin_inff(); /* pruning off */
    }

    /* waves per DP workgroup: the LUT tile is 64*(D+1) floats; keep >= 16 waves per CU */
    c->nwaves_unary = IS_UNARY_WAVES;
    c->nwaves_pairwise = IS_UNARY_WAVES;
    if (d.knob_pw_waves >= 1 && d.knob_pw_waves <= IS_UNARY_WAVES) c->nwaves_pairwise = d.knob_pw_waves;
    else d.knob_pw_waves = -1;
    if (sizeof(int) * (6 * (size_t)d.H + 3 * (size_t)d.S + 4) > 160 * 1024 ||
        isk_unary_lds_bytes(&d) > 160 * 1024 || isk_pairwise_lds_bytes(&d, c->nwaves_pairwise) > 160 * 1024 ||
        isk_prepare_lds_bytes(&d) > 160 * 1024 || isk_phase2_lds_bytes(&d) > 64 * 1024 ||
        isk_phase2s_lds_bytes(&d) > 64 * 1024 || /* (both far below: no attribute is set for them) */
        sizeof(int) * (size_t)d.C * IS_INSTANCE_CLASSES + 16 > 160 * 1024)
        return fail_arg("shape needs more than 160 KiB of LDS per workgroup");

    const size_t H = d.H, C = d.C, D = d.D, B = max_batch;
    size_t total = 0;
#define ALLOC(ptr, bytes)                                     \
    do {                                                      \
        HIP_TRY(hipMalloc((void**)&(ptr), (bytes)));          \
        total += (bytes);                                     \
    } while (0)
    ALLOC(c->d_obj_cost_lut, sizeof(float) * D * D);
    ALLOC(c->d_odr, sizeof(float) * D);
    ALLOC(c->d_rcp, sizeof(float) * (H + 1));
    ALLOC(c->d_col_flags, sizeof(int) * B * C);
    ALLOC(c->d_prune, sizeof(PruneRec) * B * C);
    ALLOC(c->d_n_generic, sizeof(int));
    HIP_TRY(hipMemset(c->d_n_generic, 0, sizeof(int))); /* later calls: k_backtrace clears it */
    c->stage_off_inst = sizeof(float) * B * 3 * H; /* (H is a multiple of 8: 8-byte aligned) */
    c->stage_off_vhor = c->stage_off_inst + sizeof(is_instance_buffers) * B;
    c->stage_bytes = c->stage_off_vhor + sizeof(int) * B;
    ALLOC(c->d_stage, c->stage_bytes);
    c->d_ground = (float*)c->d_stage;
    c->d_inst_tbl = (is_instance_buffers*)(c->d_stage + c->stage_off_inst);
    c->d_vhor = (int*)(c->d_stage + c->stage_off_vhor);
    ALLOC(c->d_recs, sizeof(RowRec) * B * C * (H + 1));
    ALLOC(c->d_lutT, sizeof(float) * B * C * (H + 1) * D);
    ALLOC(c->d_priors, sizeof(PriorRec) * B * H);
    ALLOC(c->d_steps, (size_t)64 * B * C * H);
    /* phase 1 may use up to IS_PW_MAX_SPLIT workgroups per column while columns are few */
    const size_t part_slots = (B * C > (size_t)IS_PW_SPLIT_TARGET_WGS ? B * C : (size_t)IS_PW_SPLIT_TARGET_WGS) +
                              (size_t)IS_PW_MAX_SPLIT;
    ALLOC(c->d_part_cost, sizeof(float) * part_slots * 3 * 64);
    ALLOC(c->d_part_idx, sizeof(int) * part_slots * 3 * 64);
    ALLOC(c->d_sv, sizeof(float) * B * C * 2 * (H + 1));
    ALLOC(c->d_t8row, sizeof(float) * B * C * H);
    ALLOC(c->dp.lut_ready, sizeof(int) * B * C);
    HIP_TRY(hipMemset(c->dp.lut_ready, 0, sizeof(int) * B * C));
    ALLOC(c->dp.lutf_bad, sizeof(int));
    HIP_TRY(hipMemset(c->dp.lutf_bad, 0, sizeof(int)));
    HIP_TRY(hipHostMalloc((void**)&c->h_lutf_repairs, sizeof(int), hipHostMallocMapped));
    *c->h_lutf_repairs = 0;
    HIP_TRY(hipHostGetDevicePointer((void**)&c->dp.lutf_repairs, c->h_lutf_repairs, 0));
    ALLOC(c->dp.win_lo, sizeof(int) * B * C * (size_t)c->dp.ntiles);
    HIP_TRY(hipMemset(c->dp.win_lo, 0, sizeof(int) * B * C * (size_t)c->dp.ntiles));
    ALLOC(c->d_blksum, sizeof(float) * B * C * ((size_t)d.ntiles * IS_QPT + 1) * 24);
    ALLOC(c->d_cost_table, sizeof(float) * B * C * H * 3);
    ALLOC(c->d_index_table, sizeof(int32_t) * B * C * H * 3);
    ALLOC(c->d_cluster_scratch, sizeof(int32_t) * B * IS_INSTANCE_CLASSES * 2 * C * (size_t)d.S);
    ALLOC(c->d_inst_cnt, sizeof(int) * B * C * IS_INSTANCE_CLASSES);
    ALLOC(c->d_counters, sizeof(unsigned long long) * IS_CNT_N);
#undef ALLOC
    c->scratch_bytes = total;
    for (int i = 0; i < IS_STAGE_SLOTS; i++) {
        HIP_TRY(hipHostMalloc((void**)&c->h_stage[i], c->stage_bytes));
        memset(c->h_stage[i], 0, c->stage_bytes);
        c->h_ground_pinned[i] = (float*)c->h_stage[i];
        c->h_inst_pinned[i] = (is_instance_buffers*)(c->h_stage[i] + c->stage_off_inst);
        c->h_vhor_pinned[i] = (int*)(c->h_stage[i] + c->stage_off_vhor);
        HIP_TRY(hipEventCreateWithFlags(&c->staging_free[i], hipEventDisableTiming));
    }
    for (int i = 0; i < IS_AUX_STREAMS; i++) {
        HIP_TRY(hipStreamCreateWithFlags(&c->aux_streams[i], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_joins[i], hipEventDisableTiming));
    }
    c->aux_stream = c->aux_streams[0];
    HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    for (int i = 0; i < 4; i++) HIP_TRY(hipEventCreate(&c->ev[i]));

    {
        /* device copy is TRANSPOSED ([dis][fn]): a wave of 64 consecutive fn reads one row */
        float* t = (float*)malloc(sizeof(float) * D * D);
        for (size_t fn = 0; fn < D; fn++)
            for (size_t dis = 0; dis < D; dis++) t[dis * D + fn] = obj_cost_lut[fn * D + dis];
        hipError_t e = hipMemcpy(c->d_obj_cost_lut, t, sizeof(float) * D * D, hipMemcpyHostToDevice);
        free(t);
        HIP_TRY(e);
    }
    HIP_TRY(hipMemcpy(c->d_odr, obj_disparity_range, sizeof(float) * D, hipMemcpyHostToDevice));
    {
        /* inverse_height = (float)(1./(vT+1-vB)) (StixelsKernels.cu:485, 608) doubles as the
         * reciprocal of the exact-division trick; both definitions must agree bit for bit */
        float* t = (float*)malloc(sizeof(float) * (H + 1));
        t[0] = 0.0f;
        bool same = true;
        for (size_t h = 1; h <= H; h++) {
            const float inverse_height = (float)(1. / (double)h);
            t[h] = 1.0f / (float)h;
            same = same && (t[h] == inverse_height);
        }
        hipError_t e = hipMemcpy(c->d_rcp, t, sizeof(float) * (H + 1), hipMemcpyHostToDevice);
        free(t);
        HIP_TRY(e);
        if (!same) return fail_arg("internal: (float)(1./h) != 1.0f/h for some h <= rows");
    }
    HIP_TRY(isk_set_lds_prepare(&d));
    HIP_TRY(isk_set_lds_unary(&d));
    HIP_TRY(isk_set_lds_pairwise(&d, c->nwaves_pairwise));
    HIP_TRY(isk_set_lds_backtrace(&d));
    if (debug)
        fprintf(stderr, "[is_core] unary DP: %d waves/WG, %zu B LDS/WG, occupancy API: %d WG/CU\n",
                c->nwaves_unary, isk_unary_lds_bytes(&d), isk_debug_occupancy(&d, c->nwaves_unary));
    return IS_OK;
}

int is_ctx_destroy(is_ctx* c) {
    if (!c) return IS_OK;
    DeviceScope scope(c->device);
    (void)hipDeviceSynchronize();
    if (c->graph_cache) {
        for (int i = 0; i < IS_GRAPH_ENTRIES; i++)
            if (c->graph_cache[i].valid) (void)hipGraphExecDestroy(c->graph_cache[i].exec);
        free(c->graph_cache);
    }
    (void)hipFree(c->d_obj_cost_lut); (void)hipFree(c->d_odr); (void)hipFree(c->d_rcp); (void)hipFree(c->d_col_flags); (void)hipFree(c->d_prune); (void)hipFree(c->d_n_generic); (void)hipFree(c->d_stage);
    (void)hipFree(c->d_recs); (void)hipFree(c->d_lutT); (void)hipFree(c->d_priors); (void)hipFree(c->d_steps); (void)hipFree(c->d_part_cost); (void)hipFree(c->d_part_idx); (void)hipFree(c->d_sv); (void)hipFree(c->d_blksum); (void)hipFree(c->d_t8row); (void)hipFree(c->dp.win_lo); (void)hipFree(c->dp.lut_ready); (void)hipFree(c->dp.lutf_bad);
    if (c->h_lutf_repairs) (void)hipHostFree(c->h_lutf_repairs);
    (void)hipFree(c->d_cost_table); (void)hipFree(c->d_index_table); (void)hipFree(c->d_cluster_scratch);
    (void)hipFree(c->d_inst_cnt); (void)hipFree(c->d_counters);
    for (int i = 0; i < IS_STAGE_SLOTS; i++) {
        if (c->h_stage[i]) (void)hipHostFree(c->h_stage[i]);
        if (c->staging_free[i]) (void)hipEventDestroy(c->staging_free[i]);
    }
    for (int i = 0; i < IS_AUX_STREAMS; i++) {
        if (c->aux_streams[i]) (void)hipStreamDestroy(c->aux_streams[i]);
        if (c->ev_joins[i]) (void)hipEventDestroy(c->ev_joins[i]);
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    for (int i = 0; i < 4; i++)
        if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    free(c);
    return IS_OK;
}

int is_join_columns(is_ctx* c, const float* d_big, int full_cols, int median_join, float* d_joined,
                    int n_images, void* stream) {
    if (!c || !d_big || !d_joined) return fail_arg("null pointer");
    if (n_images < 1) return fail_arg("n_images < 1");
    const is_stixel_params& p = c->params;
    if (p.column_step > 16) return fail_arg("column_step > 16");
    if (p.width_margin + p.cols * p.column_step > full_cols)
        return fail_arg("full_cols smaller than width_margin + realcols*column_step");
    ON_CTX_DEVICE(c);
    HIP_TRY(isk_launch_join(d_big, d_joined, p.rows, full_cols, p.cols, p.column_step,
                            p.width_margin, median_join, p.invalid_disparity, n_images,
                            (hipStream_t)stream));
    return IS_OK;
}

int is_cluster_instances(is_ctx* c, const is_instance_buffers* ib, void* stream) {
    if (!c || !ib) return fail_arg("null pointer");
    if (!ib->d_labels || !ib->d_centerofmass || !ib->d_core_candidates || !ib->d_instances_per_class)
        return fail_arg("d_labels, d_centerofmass, d_core_candidates and d_instances_per_class are required");
    ON_CTX_DEVICE(c);
    /* (the scratch of image slot 0: calls on one context must be stream-ordered, see the header) */
    HIP_TRY(isk_launch_cluster(c->dp.C * c->dp.S, c->params.clustering_eps, c->params.clustering_min_pts,
                               1, nullptr, ib, c->d_cluster_scratch, (hipStream_t)stream));
    return IS_OK;
}

int is_pack_sections(const is_section* d_sections, int n_columns, int max_sections, int32_t* d_counts,
                     int32_t* d_offsets, is_section* d_packed, void* stream) {
    if (!d_sections || !d_counts || !d_offsets || !d_packed) return fail_arg("null pointer");
    if (n_columns < 1 || max_sections < 2) return fail_arg("empty shape");
    if ((((uintptr_t)d_sections) | ((uintptr_t)d_packed)) & 15) return fail_arg("section arrays must be 16-byte aligned");
    HIP_TRY(isk_launch_pack(d_sections, n_columns, max_sections, d_counts, d_offsets, d_packed,
                            (hipStream_t)stream));
    return IS_OK;
}

int is_unpack_sections(const int32_t* d_counts, int32_t* d_offsets, const is_section* d_packed,
                       int n_columns, int max_sections, is_section* d_sections, void* stream) {
    if (!d_sections || !d_counts || !d_offsets || !d_packed) return fail_arg("null pointer");
    if (n_columns < 1 || max_sections < 2) return fail_arg("empty shape");
    if ((((uintptr_t)d_sections) | ((uintptr_t)d_packed)) & 15) return fail_arg("section arrays must be 16-byte aligned");
    HIP_TRY(isk_launch_unpack(d_counts, d_offsets, d_packed, n_columns, max_sections, d_sections,
                              (hipStream_t)stream));
    return IS_OK;
}

int is_flip_and_pad(const float* d_cnn_out, int32_t* d_segmentation, int n_images, int channels,
                    int rows8, int cols8, int rows_power2_segmentation, void* stream) {
    if (!d_cnn_out || !d_segmentation) return fail_arg("null pointer");
    if (n_images < 1 || channels < 1 || rows8 < 1 || cols8 < 1) return fail_arg("empty shape");
    if (rows_power2_segmentation < rows8 + 1 ||
        (rows_power2_segmentation & (rows_power2_segmentation - 1)) != 0)
        return fail_arg("rows_power2_segmentation must be a power of two > rows/8 (Stixels.cu:132-133)");
    HIP_TRY(isk_launch_flip_and_pad(d_cnn_out, d_segmentation, n_images, channels, rows8, cols8,
                                    rows_power2_segmentation, (hipStream_t)stream));
    return IS_OK;
}

int is_road_vdisparity(const float* d_disparity, int rows, int cols, int max_dis, float threshold,
                       int* d_vdisp, int* d_maximum, uint8_t* d_binary, void* stream) {
    if (!d_disparity || !d_vdisp || !d_maximum || !d_binary) return fail_arg("null pointer");
    if (rows < 1 || cols < 1 || max_dis < 1 || max_dis > 16384) return fail_arg("bad shape");
    HIP_TRY(isk_launch_vdisparity(d_disparity, d_vdisp, d_maximum, d_binary, rows, cols, max_dis,
                                  threshold, (hipStream_t)stream));
    return IS_OK;
}

static_assert(IS_CNT_N == IS_EVAL_COUNTERS, "is_device.h and instance_stixels_core.h disagree on the counter array");

int is_set_eval_counters(is_ctx* c, int enabled) {
    if (!c) return fail_arg("null ctx");
    ON_CTX_DEVICE(c);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(c->d_counters, 0, sizeof(unsigned long long) * IS_CNT_N));
    c->counting = enabled != 0;
    return IS_OK;
}

int is_get_eval_counters(is_ctx* c, unsigned long long* out, int n) {
    if (!c || !out) return fail_arg("null pointer");
    if (n < 1 || n > IS_CNT_N) return fail_arg("n outside [1, IS_EVAL_COUNTERS]");
    ON_CTX_DEVICE(c);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, c->d_counters, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost));
    return IS_OK;
}

int is_debug_read_object_lut(is_ctx* c, int column, float* h_out) {
    if (!c || !h_out) return fail_arg("null pointer");
    if (column < 0 || column >= c->max_batch * c->dp.C) return fail_arg("column outside the context's scratch");
    ON_CTX_DEVICE(c);
    HIP_TRY(hipDeviceSynchronize());
    const size_t n = ((size_t)c->dp.H + 1) * c->dp.D;
    HIP_TRY(hipMemcpy(h_out, c->d_lutT + (size_t)column * n, sizeof(float) * n, hipMemcpyDeviceToHost));
    return IS_OK;
}

int is_debug_lut_fused_state(is_ctx* c, int* repaired) {
    if (!c || !repaired) return fail_arg("null pointer");
    ON_CTX_DEVICE(c);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(repaired, c->dp.lutf_bad, sizeof(int), hipMemcpyDeviceToHost));
    return IS_OK;
}

int is_lut_fused_repairs(is_ctx* c, int* calls_repaired) {
    if (!c || !calls_repaired) return fail_arg("null pointer");
    ON_CTX_DEVICE(c);
    HIP_TRY(hipDeviceSynchronize());
    *calls_repaired = c->h_lutf_repairs ? *(volatile int*)c->h_lutf_repairs : 0;
    return IS_OK;
}

int is_debug_read_block_summaries(is_ctx* c, int column, float* h_out, int cap_floats, int* n_blocks) {
    if (!c || !h_out || !n_blocks) return fail_arg("null pointer");
    if (column < 0 || column >= c->max_batch * c->dp.C) return fail_arg("column outside the context's scratch");
    const int nb = c->dp.ntiles * IS_QPT + 1;
    if (cap_floats < nb * 24) return fail_arg("h_out too small");
    ON_CTX_DEVICE(c);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(h_out, c->d_blksum + (size_t)column * nb * 24, sizeof(float) * nb * 24, hipMemcpyDeviceToHost));
    *n_blocks = nb;
    return IS_OK;
}

int is_set_kernel_timing(is_ctx* c, int enabled) {
    if (!c) return fail_arg("null ctx");
    c->timing = enabled != 0;
    c->ev_valid = false;
    return IS_OK;
}

int is_get_kernel_times_ms(is_ctx* c, float* prepare_ms, float* dp_ms, float* backtrace_ms) {
    if (!c) return fail_arg("null ctx");
    if (!c->timing || !c->ev_valid) return fail_arg("kernel timing not enabled / no call recorded");
    HIP_TRY(hipEventSynchronize(c->ev[3]));
    float a = 0, b = 0, d = 0;
    HIP_TRY(hipEventElapsedTime(&a, c->ev[0], c->ev[1]));
    HIP_TRY(hipEventElapsedTime(&b, c->ev[1], c->ev[2]));
    HIP_TRY(hipEventElapsedTime(&d, c->ev[2], c->ev[3]));
    if (prepare_ms) *prepare_ms = a;
    if (dp_ms) *dp_ms = b;
    if (backtrace_ms) *backtrace_ms = d;
    return IS_OK;
}

/* fn windows of the DP kernels (is_device.h, IS_P1_WIN) for the tiles that start below the horizon of every image
 * of the call: ground and what stands on it span few disparities within 64 rows, while a tile above the
 * horizon mixes sky (d ~ 0) with objects of any disparity -- measured: 3.6 % of the steps of tile 7 read
 * outside the window, 22-35 % of tiles 12-13, and a step with a lane outside pays a memory round trip.
 * The split decides launch geometry only (workgroup shapes, which kernel instantiation a tile runs), never
 * results; a hipGraph replays the split it was captured with, so it is part of the graph cache's key. */
static int call_win_tiles(const DevParams& P, const int* h_vhor, int n_images, int pairwise) {
    int vmin = P.H;
    for (int i = 0; i < n_images; i++) vmin = h_vhor[i] < vmin ? h_vhor[i] : vmin;
    int w = (IS_P1_WINDOWED(P.D) && P.win_lo != nullptr && vmin > 0) ? (vmin + IS_TILE - 1) / IS_TILE : 0;
    /* the unary kernel windows EVERY tile: a lane outside costs it an L2 gather (2.8 % of its steps on the
     * synthetic scene), not the HBM round trip on a latency-bound chain it costs phase 1 -- measured at
     * batch 64: 9 windowed tiles 7450, all 16: 7760 frames/s (pairwise 3780 / 3810, but its unpruned
     * floor 1720 / 1630) */
    if (!pairwise && IS_P1_WINDOWED(P.D) && P.win_lo != nullptr) w = P.ntiles;
    if (P.knob_win_tiles >= 0) w = P.knob_win_tiles; /* (experiments, tests) */
    return w;
}

/* Everything is_compute queues on `stream` behind the host-side staging of slot `slot`.
 * `capturing`: the calls are being recorded into a hipGraph -- no timing events, no staging
 * event (the caller records it behind the graph launch). */
static int compute_enqueue(is_ctx* c, const float* d_joined, const int32_t* d_seg, int pairwise, int n_images,
                           is_section* d_sections, const is_instance_buffers* instances, float* d_cost_table,
                           int32_t* d_index_table, hipStream_t stream, int slot, bool capturing) {
    const DevParams& P = c->dp;
    const size_t H = P.H;
    const int ncols = n_images * P.C;
    const bool timing = c->timing && !capturing;
    const bool one_copy = n_images == c->max_batch;
    if (one_copy) {
        HIP_TRY(hipMemcpyAsync(c->d_stage, c->h_stage[slot], c->stage_bytes, hipMemcpyHostToDevice, stream));
    } else {
        HIP_TRY(hipMemcpyAsync(c->d_ground, c->h_ground_pinned[slot], sizeof(float) * n_images * 3 * H,
                               hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(c->d_vhor, c->h_vhor_pinned[slot], sizeof(int) * n_images,
                               hipMemcpyHostToDevice, stream));
    }
    bool want_inst = false, want_labels = false;
    if (instances)
        for (int i = 0; i < n_images; i++) {
            const is_instance_buffers& ib = instances[i];
            want_inst = want_inst || ib.d_centerofmass || ib.d_indices || ib.d_core_candidates ||
                        ib.d_instances_per_class;
            want_labels = want_labels || ib.d_labels;
        }
    if (want_inst && !one_copy) /* the per-image output pointers travel through the pinned staging slot of this call */
        HIP_TRY(hipMemcpyAsync(c->d_inst_tbl, c->h_inst_pinned[slot], sizeof(is_instance_buffers) * n_images,
                               hipMemcpyHostToDevice, stream));
    if (!capturing) HIP_TRY(hipEventRecord(c->staging_free[slot], stream));

    float* ct = d_cost_table ? d_cost_table : c->d_cost_table;
    int32_t* it = d_index_table ? d_index_table : c->d_index_table;

    DevParams Pw = P; /* (+ this call's windowed / classic tile split and the form of lutT) */
    Pw.win_tiles = call_win_tiles(P, c->h_vhor_pinned[slot], n_images, pairwise);
    Pw.lut_carry = (!pairwise && isk_unary_uses_carry(&Pw, ncols)) ? 1 : 0;
    Pw.lut_fused = (!pairwise && !capturing) ? isk_unary_uses_fused_lut(&Pw, ncols) : 0;
    /* a hand-over of this context has been distrusted before (another dispatcher, a partition mode, a CU mask): the
     * fused launch stays off unless IS_LUT_FUSED asks for it by value -- a repaired call costs 2.8 x an ordinary one */
    if (Pw.lut_fused && (P.knob_lut_fused < 0 || P.knob_lut_fused == 3) && c->h_lutf_repairs && *(volatile int*)c->h_lutf_repairs > 0) Pw.lut_fused = 0;
    if (timing) HIP_TRY(hipEventRecord(c->ev[0], stream));
    /* (d_n_generic is zero here: cleared at creation and by k_backtrace at the end of every call) */
    HIP_TRY(isk_launch_prepare(&Pw, ncols, d_joined, d_seg, c->d_ground, c->d_vhor,
                               c->d_obj_cost_lut, c->d_recs, c->d_lutT, c->d_col_flags, c->d_sv,
                               c->d_prune, c->d_n_generic, stream, c->aux_stream, c->ev_fork, c->ev_join));
    if (pairwise) HIP_TRY(isk_launch_priors(&P, c->d_ground, c->d_priors, n_images, stream));
    if (timing) HIP_TRY(hipEventRecord(c->ev[1], stream));
    if (pairwise)
        HIP_TRY(isk_launch_dp_pairwise(&Pw, ncols, c->nwaves_pairwise, c->d_recs, c->d_lutT,
                                       d_joined, c->d_priors, c->d_odr, c->d_rcp, c->d_sv, c->d_vhor,
                                       c->d_col_flags, c->d_prune, c->d_steps, c->d_part_cost,
                                       c->d_part_idx, ct, it, c->counting ? c->d_counters : nullptr,
                                       c->d_obj_cost_lut, c->d_n_generic, c->d_blksum, c->d_t8row, stream, c->aux_streams,
                                       IS_AUX_STREAMS, c->ev_fork, c->ev_joins));
    else
        HIP_TRY(isk_launch_dp_unary(&Pw, ncols, c->nwaves_unary, c->d_recs, c->d_lutT, c->d_rcp,
                                    c->d_vhor, c->d_col_flags, c->d_prune, ct, it, c->d_n_generic,
                                    c->counting ? c->d_counters : nullptr, d_joined, c->d_obj_cost_lut,
                                    stream));
    if (timing) HIP_TRY(hipEventRecord(c->ev[2], stream));
    HIP_TRY(isk_launch_backtrace(&P, ncols, pairwise ? 1 : 0, c->d_recs, ct, it, c->d_col_flags,
                                 d_sections, want_inst ? c->d_inst_cnt : nullptr, c->d_n_generic, stream));
    if (want_inst) {
        /* the instance candidates (StixelsKernels.cu:926-942) and their clustering
         * (Stixels::ClusterInstances, Stixels.cu:613) of the WHOLE batch: two launches */
        HIP_TRY(isk_launch_compact(&P, n_images, d_sections, c->d_inst_cnt, c->d_inst_tbl, stream));
        if (want_labels)
            HIP_TRY(isk_launch_cluster(P.C * P.S, c->params.clustering_eps, c->params.clustering_min_pts,
                                       n_images, c->d_inst_tbl, nullptr, c->d_cluster_scratch, stream));
    }
    if (timing) {
        HIP_TRY(hipEventRecord(c->ev[3], stream));
        c->ev_valid = true;
    }
    return IS_OK;
}

/* ---- hipGraph replay of small calls (opt-in: IS_GRAPH=1) ----------------------------------------
 * A single frame is ~15 queue operations (copies, a memset, 6-38 kernels) of a few microseconds
 * each.  The whole sequence of a call only depends on the pointers and sizes the caller passes, so
 * calls of up to IS_GRAPH_MAX_IMAGES images can be captured once per distinct argument set and
 * replayed with one hipGraphLaunch; the per-frame host data (ground model, horizon) still travels
 * through the pinned staging slot the graph's copy nodes read.  MEASURED (round 3, ROCm 7.2, one
 * 1024x2048 frame through Stixels::Compute): 0.455 ms replayed against 0.400 ms launched eagerly
 * (pairwise 1.92 vs 1.91 ms) -- the gaps between dependent dispatches are the GPU's, not the
 * host's, and a replay adds its own launch cost; so the path is off unless IS_GRAPH is set
 * (bit-exact on the whole GPU suite).  Needs a real stream: the legacy NULL stream cannot be captured.
 * A cached graph owns ONE pinned staging slot (its copy nodes read it): every replay waits on the host
 * for the previous replay's staging event before it refills the slot, i.e. replays of one graph
 * serialise on that copy. */
static bool graph_matches(const is_graph_entry& e, const float* d_joined, const int32_t* d_seg, int pairwise,
                          int n_images, const is_section* d_sections, const is_instance_buffers* instances,
                          const float* ct, const int32_t* it, int win_tiles) {
    if (!e.valid || e.win_tiles != win_tiles || e.joined != d_joined || e.seg != d_seg || e.sections != d_sections || e.ct != ct ||
        e.it != it || e.pairwise != pairwise || e.n_images != n_images || e.n_inst != (instances ? n_images : 0))
        return false;
    return !instances || memcmp(e.inst, instances, sizeof(is_instance_buffers) * n_images) == 0;
}

int is_compute(is_ctx* c, const float* d_joined, const int32_t* d_seg, const float* h_gf,
               const float* h_ng, const float* h_is2, const int* h_vhor, int pairwise, int n_images,
               is_section* d_sections, const is_instance_buffers* instances, float* d_cost_table,
               int32_t* d_index_table, void* stream_) {
    if (!c || !d_joined || !d_seg || !h_gf || !h_ng || !h_is2 || !h_vhor || !d_sections)
        return fail_arg("null pointer");
    if (n_images < 1 || n_images > c->max_batch) return fail_arg("n_images outside [1, max_batch]");
    if ((((uintptr_t)d_joined) | ((uintptr_t)d_seg)) & 15)
        return fail_arg("d_joined / d_segmentation must be 16-byte aligned (vector loads)");
    if (instances)
        for (int i = 0; i < n_images; i++)
            if (instances[i].d_labels && (!instances[i].d_centerofmass || !instances[i].d_core_candidates ||
                                          !instances[i].d_instances_per_class))
                return fail_arg("d_labels needs d_centerofmass, d_core_candidates and d_instances_per_class");
    ON_CTX_DEVICE(c);
    hipStream_t stream = (hipStream_t)stream_;
    const size_t H = c->dp.H;

    /* a cached graph of exactly this call?  (its staging slot is part of the graph) */
    const bool graph_ok = c->graphs && stream != nullptr && n_images <= IS_GRAPH_MAX_IMAGES && !c->timing &&
                          !c->counting;
    is_graph_entry* ge = nullptr;
    const int win_tiles = call_win_tiles(c->dp, h_vhor, n_images, pairwise);
    if (graph_ok)
        for (int i = 0; i < IS_GRAPH_ENTRIES; i++)
            if (graph_matches(c->graph_cache[i], d_joined, d_seg, pairwise, n_images, d_sections, instances,
                              d_cost_table, d_index_table, win_tiles))
                ge = &c->graph_cache[i];

    /* stage the per-frame ground model (the reference does 3 blocking cudaMemcpy per frame,
     * Stixels.cu:479-493): pinned + async here, through a ring of slots each guarded by an event */
    int slot;
    if (ge) {
        slot = ge->slot;
    } else {
        slot = c->stage_next;
        c->stage_next = (slot + 1) % IS_STAGE_SLOTS;
    }
    if (c->staging_pending[slot]) HIP_TRY(hipEventSynchronize(c->staging_free[slot]));
    for (int i = 0; i < n_images; i++) {
        float* dst = c->h_ground_pinned[slot] + (size_t)i * 3 * H;
        memcpy(dst, h_gf + (size_t)i * H, sizeof(float) * H);
        memcpy(dst + H, h_ng + (size_t)i * H, sizeof(float) * H);
        memcpy(dst + 2 * H, h_is2 + (size_t)i * H, sizeof(float) * H);
        c->h_vhor_pinned[slot][i] = h_vhor[i];
    }
    /* (no instances: zeros, so that the one-copy path of a full batch never leaves pointers of an
     * older call -- possibly freed since -- in d_inst_tbl) */
    if (instances) memcpy(c->h_inst_pinned[slot], instances, sizeof(is_instance_buffers) * n_images);
    else memset(c->h_inst_pinned[slot], 0, sizeof(is_instance_buffers) * n_images);

    if (graph_ok && !ge) { /* first call with these arguments: record it */
        is_graph_entry* victim = &c->graph_cache[0];
        for (int i = 1; i < IS_GRAPH_ENTRIES; i++)
            if (!c->graph_cache[i].valid || (victim->valid && c->graph_cache[i].last_use < victim->last_use))
                victim = &c->graph_cache[i];
        if (victim->valid) {
            (void)hipGraphExecDestroy(victim->exec);
            victim->valid = false;
        }
        hipGraph_t graph = nullptr;
        hipError_t e = hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed);
        if (e == hipSuccess) {
            const int rc = compute_enqueue(c, d_joined, d_seg, pairwise, n_images, d_sections, instances,
                                           d_cost_table, d_index_table, stream, slot, true);
            e = hipStreamEndCapture(stream, &graph);
            if (rc == IS_OK && e == hipSuccess && graph) {
                hipGraphExec_t exec = nullptr;
                e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
                if (e == hipSuccess) {
                    victim->valid = true;
                    victim->joined = d_joined; victim->seg = d_seg; victim->sections = d_sections;
                    victim->ct = d_cost_table; victim->it = d_index_table;
                    victim->pairwise = pairwise; victim->n_images = n_images;
                    victim->win_tiles = win_tiles;
                    victim->n_inst = instances ? n_images : 0;
                    if (instances) memcpy(victim->inst, instances, sizeof(is_instance_buffers) * n_images);
                    victim->slot = slot;
                    victim->exec = exec;
                    ge = victim;
                }
            }
            if (graph) (void)hipGraphDestroy(graph);
        }
        if (!ge) { /* capture is not available here: never try again on this context */
            (void)hipGetLastError();
            c->graphs = false;
        }
    }
    if (ge) {
        ge->last_use = ++c->graph_clock;
        HIP_TRY(hipGraphLaunch(ge->exec, stream));
        HIP_TRY(hipEventRecord(c->staging_free[slot], stream));
        c->staging_pending[slot] = true;
        return IS_OK;
    }
    c->staging_pending[slot] = true;
    const int rc = compute_enqueue(c, d_joined, d_seg, pairwise, n_images, d_sections, instances, d_cost_table,
                                   d_index_table, stream, slot, false);
    /* Invariant the early-outs of k_dp_unary / k_pw_phase2_generic rely on: d_n_generic is zero
     * between calls (k_prepare counts the generic columns of a call, block 0 of k_backtrace clears
     * the counter at its end).  A call that failed half way may have counted without clearing. */
    if (rc != IS_OK) (void)hipMemsetAsync(c->d_n_generic, 0, sizeof(int), stream);
    return rc;
}
