/*
 * stixels_oracle.c -- CPU restatement of the reference's column-DP hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see stixels_oracle.h for the rules and the parity-pin status:
 * "parity unpinned" beyond the scan / column-join known answers of the reference's own
 * disabled unit tests).
 *
 * The restatement follows the reference's SIMT structure literally: every region between
 * two __syncthreads() of the CUDA kernel becomes a loop over the block's threads, every
 * formula keeps the reference's operand order, types (fp32 / int32 / int64 / fp64
 * temporaries) and comparison directions.  All `file:line` citations are relative to
 * /root/reference/InstanceStixels/ (src/ and include/InstanceStixels/).
 *
 * Canonical numerics: IEEE fp32, round-to-nearest-even, no FMA contraction,
 * `logf` inside the kernel = is_logf (include/is_numerics.h); host precompute uses libm as
 * the reference host code does.  Build with: -O2 -ffp-contract=off -fwrapv -fno-fast-math.
 */
#include "stixels_oracle.h"
#include "is_numerics.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_INF (__builtin_inff())     /* MAX_LOGPROB = CUDART_INF_F, configuration.h:29 */
#define LOG_LUT_SIZE 1000000           /* configuration.h:30 */
#define DSF IS_DOWNSAMPLE_FACTOR       /* configuration.h:31 */
#define WARP 32                        /* util.h:25 */
#define PIFLOAT 3.1416f                /* Stixels.hpp:38 */

float orc_logf(float x) { return is_logf(x); }

/* RoadEstimationKernels.cu:25-60, one loop iteration per CUDA thread */
int orc_road_vdisparity(const float* d_disparity, int rows, int cols, int max_dis, float threshold,
                        int* d_vDisp, uint8_t* d_out) {
    memset(d_vDisp, 0, sizeof(int) * (size_t)rows * max_dis);
    for (int idx = 0; idx < cols * rows; idx++) { /* ComputeHistogram, :25-39 */
        const int row = idx / cols;
        const float d = d_disparity[idx];
        if (d != 0) {
            int col = (int)d;
            d_vDisp[row * max_dis + col] += 1;
        }
    }
    int maximum = 0;
    for (int idx = 0; idx < max_dis * rows; idx++) /* ComputeMaximum, :42-50 */
        if (d_vDisp[idx] > maximum) maximum = d_vDisp[idx];
    for (int idx = 0; idx < max_dis * rows; idx++) { /* ComputeBinaryImage, :52-60 */
        const float p = (float)d_vDisp[idx];
        d_out[idx] = (p > maximum * threshold) ? 255 : 0;
    }
    return maximum;
}

/* FlipAndPad.forward, tools/CNN_training/models/wrappers.py:44-61 */
void orc_flip_and_pad(const float* in, int32_t* out, int CH, int Hs, int Ws, int P2S) {
    for (int w = 0; w < Ws; w++)            /* x.permute(0,3,1,2): [Ws][CH][Hs] */
        for (int c = 0; c < CH; c++)
            for (int k = 0; k < P2S; k++) {
                int32_t v = 0;              /* F.pad(..., value=0) on the right of the last dim */
                if (k < Hs) {
                    float x = in[((size_t)c * Hs + (Hs - 1 - k)) * Ws + w]; /* index_select flip */
                    x *= 8;                 /* x *= 8 */
                    v = (int32_t)x;         /* x.int(): truncation toward zero */
                }
                out[((size_t)w * CH + c) * P2S + k] = v;
            }
}

/* ------------------------------------------------------------------------------------ */
/* Host side: Stixels::SetConfig / Initialize / PrecomputeGround                         */
/* ------------------------------------------------------------------------------------ */

void orc_default_config(orc_config* c) { /* types.h:30-141 */
    memset(c, 0, sizeof(*c));
    c->rows = -1; c->cols = -1; c->max_dis = -1;
    c->invalid_disparity = -1.0f;
    c->eps = -1; c->min_pts = -1; c->size_filter = -1;
    c->n_semantic_classes = -1; c->n_offset_channels = -1;
    c->prior_weight = -1; c->segmentation_weight = -1;
    c->instance_weight = -1; c->disparity_weight = -1;
    c->column_step = -1;
    c->focal = -1; c->baseline = -1; c->camera_center_x = -1; c->camera_center_y = -1;
    c->sigma_disparity_object = 1.0f; c->sigma_disparity_ground = 2.0f; c->sigma_sky = 0.1f;
    c->pout = 0.15f; c->pout_sky = 0.4f; c->pord = 0.2f; c->pgrav = 0.1f; c->pblg = 0.04f;
    c->pground_given_nexist = 0.28; c->pobject_given_nexist = 0.44;
    c->psky_given_nexist = 0.28;
    c->pnexist_dis = 0.25f;
    c->pground = 1.0f / 3.0f; c->pobject = 1.0f / 3.0f; c->psky = 1.0f / 3.0f;
    c->width_margin = 0;
    c->sigma_camera_tilt = 0.05f; c->sigma_camera_height = 0.05f;
    c->median_join = 0; c->epsilon = 3.0f; c->range_objects_z = 10.20f;
}

typedef struct { /* the m_* members of class Stixels that the precompute needs */
    float pout, pout_sky, pnex_g, pnex_o, pnex_s;
    float focal, baseline, sigma_camera_tilt, sigma_camera_height;
    float max_disf, sigma_obj, sigma_gnd, sigma_sky, range_z, invalid;
    int max_dis, rows;
    float* log_lut;
} host_state;

static float* build_log_lut(void) { /* Stixels.cu:79-84 */
    float* lut = (float*)malloc(sizeof(float) * (LOG_LUT_SIZE + 1));
    for (int i = 0; i < LOG_LUT_SIZE; i++) {
        const float log_res = (float)i / ((float)LOG_LUT_SIZE);
        lut[i] = logf(log_res);
    }
    lut[LOG_LUT_SIZE] = 0.0f;
    return lut;
}

static float fast_log(const host_state* s, float v) { /* Stixels.cu:786-788 */
    return s->log_lut[(int)((v)*LOG_LUT_SIZE + 0.5f)];
}

static void host_state_from_config(const orc_config* c, host_state* s) {
    /* SetProbabilities, Stixels.cu:361-373 */
    s->pout = c->pout;
    s->pout_sky = c->pout_sky;
    s->pnex_g = (c->pground_given_nexist * c->pnexist_dis) / c->pground;
    s->pnex_o = (c->pobject_given_nexist * c->pnexist_dis) / c->pobject;
    s->pnex_s = (c->psky_given_nexist * c->pnexist_dis) / c->psky;
    /* SetCameraParameters, Stixels.cu:383-393 */
    s->focal = c->focal;
    s->baseline = c->baseline;
    s->sigma_camera_tilt = c->sigma_camera_tilt * (PIFLOAT) / 180.0f;
    s->sigma_camera_height = c->sigma_camera_height;
    /* SetDisparityParameters, Stixels.cu:425-437 */
    s->rows = c->rows;
    s->max_dis = c->max_dis;
    s->max_disf = (float)c->max_dis;
    s->sigma_obj = c->sigma_disparity_object;
    s->sigma_gnd = c->sigma_disparity_ground;
    s->sigma_sky = c->sigma_sky;
    s->invalid = c->invalid_disparity;
    s->range_z = c->range_objects_z;
}

int orc_host_initialize(const orc_config* c, is_stixel_params* p, float* obj_cost_lut,
                        float* obj_disparity_range) {
    /* SetConfig checks, Stixels.cu:292-313 */
    if (c->rows == -1 || c->cols == -1 || c->max_dis == -1) return -1;
    if (c->eps == -1 || c->min_pts == -1 || c->size_filter == -1) return -1;
    if (c->prior_weight == -1 || c->segmentation_weight == -1 || c->instance_weight == -1 ||
        c->disparity_weight == -1)
        return -1;
    if (c->column_step == -1 || c->focal == -1 || c->baseline == -1) return -1;

    host_state s;
    host_state_from_config(c, &s);
    s.log_lut = build_log_lut();
    memset(p, 0, sizeof(*p));

    const int realcols = (c->cols - c->width_margin) / c->column_step; /* Stixels.cu:44 */

    /* frequently used values, Stixels.cu:93-102 */
    const float max_dis_log = logf(s.max_disf);
    const float rows_log = logf((float)c->rows);
    const float puniform_sky = max_dis_log - logf(s.pout_sky);
    const float puniform = max_dis_log - logf(s.pout);
    const float pnex_s_log = -logf(s.pnex_s), nopnex_s_log = -logf(1.0f - s.pnex_s);
    const float pnex_g_log = -logf(s.pnex_g), nopnex_g_log = -logf(1.0f - s.pnex_g);
    const float pnex_o_log = -logf(s.pnex_o), nopnex_o_log = -logf(1.0f - s.pnex_o);

    /* ComputeObjectDisparityRange, Stixels.cu:111-115, 879-887 */
    for (int i = 0; i < c->max_dis; i++) {
        const float previous_mean = (float)i;
        float range_disp = 0.0f;
        if (previous_mean != 0) {
            const float pmean_plus_z = (s.baseline * s.focal / previous_mean) + s.range_z;
            range_disp = previous_mean - (s.baseline * s.focal / pmean_plus_z);
        }
        obj_disparity_range[i] = range_disp;
    }

    /* PrecomputeSky, Stixels.cu:856-865 */
    float normalization_sky, inv_sigma2_sky;
    {
        const float sigma = s.sigma_sky;
        const float pout = s.pout_sky;
        const float a_range = 0.5f * (erff(s.max_disf / (sigma * sqrtf(2.0f))) - erff(0.0f));
        normalization_sky =
            fast_log(&s, a_range) - logf((1.0f - pout) / (sigma * sqrtf(2.0f * PIFLOAT)));
        inv_sigma2_sky = 1.0f / (2.0f * sigma * sigma);
    }

    /* PrecomputeObject, Stixels.cu:819-840 */
    float* norm_obj = (float*)malloc(sizeof(float) * c->max_dis);
    float* inv_s2_obj = (float*)malloc(sizeof(float) * c->max_dis);
    for (int dis = 0; dis < c->max_dis; dis++) {
        const float fn = (float)dis;
        const float sigma_object = fn * fn * s.range_z / (s.focal * s.baseline);
        const float sigma = sqrtf(s.sigma_obj * s.sigma_obj + sigma_object * sigma_object);
        const float a_range = 0.5f * (erff((s.max_disf - fn) / (sigma * sqrtf(2.0f))) -
                                      erff((-fn) / (sigma * sqrtf(2.0f))));
        norm_obj[dis] = fast_log(&s, a_range) -
                        fast_log(&s, (1.0f - s.pout) / (sigma * sqrtf(2.0f * PIFLOAT)));
        inv_s2_obj[dis] = 1.0f / (2.0f * sigma * sigma);
    }

    /* GetDataCostObject, Stixels.cu:122-129, 842-854 */
    for (int fn = 0; fn < c->max_dis; fn++) {
        for (int dis = 0; dis < c->max_dis; dis++) {
            float data_cost = pnex_o_log;
            if (dis != (int)s.invalid) {
                const float model_diff = (float)(dis - fn);
                const float pgaussian = norm_obj[fn] + model_diff * model_diff * inv_s2_obj[fn];
                const float p_data = fminf(puniform, pgaussian);
                data_cost = p_data + nopnex_o_log;
            }
            obj_cost_lut[fn * c->max_dis + dis] = data_cost;
        }
    }
    free(norm_obj);
    free(inv_s2_obj);
    free(s.log_lut);

    /* Stixels.cu:131-133 */
    const int rows_power2 = (int)powf(2, ceilf(log2f(c->rows + 1)));
    const int rows_power2_segmentation = (int)powf(2, ceilf(log2f(c->rows / 8 + 1)));

    /* SetWeightParameters, Stixels.cu:408-423 */
    float instance_weight = 0.0;
    if (c->segmentation_weight > 1e-5) {
        instance_weight = c->instance_weight / c->segmentation_weight;
        if (c->instance_weight < 1e-8) instance_weight = 0.0;
    }

    /* m_params fill, Stixels.cu:212-245 and SetClusteringParameters :395-400 */
    p->vhor = 0;
    p->rows = c->rows;
    p->cols = realcols;
    p->max_dis = c->max_dis;
    p->invalid_disparity = c->invalid_disparity;
    p->rows_log = rows_log;
    p->pnexists_given_sky_log = pnex_s_log;
    p->normalization_sky = normalization_sky;
    p->inv_sigma2_sky = inv_sigma2_sky;
    p->puniform_sky = puniform_sky;
    p->nopnexists_given_sky_log = nopnex_s_log;
    p->pnexists_given_ground_log = pnex_g_log;
    p->puniform = puniform;
    p->nopnexists_given_ground_log = nopnex_g_log;
    p->pnexists_given_object_log = pnex_o_log;
    p->nopnexists_given_object_log = nopnex_o_log;
    p->baseline = c->baseline;
    p->focal = c->focal;
    p->range_objects_z = c->range_objects_z;
    p->pord = c->pord;
    p->epsilon = c->epsilon;
    p->pgrav = c->pgrav;
    p->pblg = c->pblg;
    p->rows_power2 = rows_power2;
    p->rows_power2_segmentation = rows_power2_segmentation;
    p->max_sections = IS_MAX_STIXELS_PER_COLUMN;
    p->max_dis_log = max_dis_log;
    p->width_margin = c->width_margin;
    p->segmentation_classes = c->n_semantic_classes;
    p->segmentation_channels = c->n_semantic_classes + c->n_offset_channels;
    p->prior_weight = c->prior_weight;
    p->disparity_weight = c->disparity_weight;
    p->segmentation_weight = c->segmentation_weight;
    p->instance_weight = instance_weight;
    p->column_step = c->column_step;
    p->clustering_eps = c->eps;
    p->clustering_min_pts = c->min_pts;
    p->clustering_size_filter = c->size_filter;
    return 0;
}

int orc_host_ground(const orc_config* c, int vhor_image, float camera_tilt, float camera_height,
                    float alpha_ground, float* ground_function, float* normalization_ground,
                    float* inv_sigma2_ground, int* vhor_lib) {
    host_state s;
    host_state_from_config(c, &s);
    s.log_lut = build_log_lut();
    const int m_vhor = c->rows - vhor_image - 1; /* SetRoadParameters, Stixels.cu:377 */
    *vhor_lib = m_vhor;

    /* PrecomputeGround, Stixels.cu:790-817 */
    const float fb = (s.focal * s.baseline) / camera_height;
    const float pout = s.pout;
    for (int v = 0; v < c->rows; v++) {
        const float fn = alpha_ground * (float)(m_vhor - v); /* GroundFunction :867-877 */
        ground_function[v] = fn;

        const float x = camera_tilt + (float)(m_vhor - v) / s.focal;
        const float sigma2_road =
            fb * fb *
            (s.sigma_camera_height * s.sigma_camera_height * x * x /
                 (camera_height * camera_height) +
             s.sigma_camera_tilt * s.sigma_camera_tilt);
        const float sigma = sqrtf(s.sigma_gnd * s.sigma_gnd + sigma2_road);

        const float a_range = 0.5f * (erff((s.max_disf - fn) / (sigma * sqrtf(2.0f))) -
                                      erff((-fn) / (sigma * sqrtf(2.0f))));

        normalization_ground[v] =
            fast_log(&s, a_range) -
            fast_log(&s, (1.0f - pout) / (sigma * sqrtf(2.0f * PIFLOAT)));
        inv_sigma2_ground[v] = 1.0f / (2.0f * sigma * sigma);
    }
    free(s.log_lut);
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* JoinColumns, StixelsKernels.cu:980-1095 (one "thread" per (row, col))                 */
/* ------------------------------------------------------------------------------------ */

static float median_of(float* tmp_row, int n) { /* selection sort, :1007-1022 / :1038-1053 */
    for (int i = 0; i < (n / 2) + 1; i++) {
        int min_idx = i;
        for (int j = i + 1; j < n; j++)
            if (tmp_row[j] < tmp_row[min_idx]) min_idx = j;
        const float tmp = tmp_row[i];
        tmp_row[i] = tmp_row[min_idx];
        tmp_row[min_idx] = tmp;
    }
    float median = tmp_row[n / 2];
    if (n % 2 == 0) median = (median + tmp_row[(n / 2) - 1]) / 2.0f;
    return median;
}

void orc_join_columns(const float* d_disparity, float* d_out, int step_size, int median,
                      int width_margin, int rows, int cols, int real_cols,
                      float invalid_disparity) {
    for (int idx = 0; idx < real_cols * rows; idx++) {
        const int row = idx / real_cols;
        const int col = idx % real_cols;
        const float* src = &d_disparity[row * cols + col * step_size + width_margin];
        float* dst = &d_out[col * rows + rows - row - 1];
        if (median) {
            float tmp_row[16];
            if (invalid_disparity >= 0) { /* :992-1028 */
                int valid_pixels = 0;
                for (int i = 0; i < step_size; i++) {
                    const float tmp = src[i];
                    if (tmp != invalid_disparity) tmp_row[valid_pixels++] = tmp;
                }
                if (valid_pixels > 0)
                    *dst = median_of(tmp_row, valid_pixels);
                else
                    *dst = invalid_disparity;
            } else { /* :1029-1055 */
                for (int i = 0; i < step_size; i++) tmp_row[i] = src[i];
                *dst = median_of(tmp_row, step_size);
            }
        } else {
            float mean = 0.0f;
            if (invalid_disparity >= 0) { /* :1068-1086 */
                int invalid = 0;
                for (int i = 0; i < step_size; i++) {
                    const float d = src[i];
                    if (d != invalid_disparity)
                        mean += d;
                    else
                        invalid++;
                }
                if (invalid != step_size)
                    *dst = mean / (step_size - invalid);
                else
                    *dst = invalid_disparity;
            } else { /* :1087-1092 */
                for (int i = 0; i < step_size; i++) mean += src[i];
                *dst = mean / step_size;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* ComputePrefixSum<T>, StixelsKernels.h:73-103                                          */
/* Each inner loop over t is one barrier-delimited region; threads t < d touch disjoint   */
/* (ai, bi) pairs, so a sequential loop is equivalent.                                    */
/* ------------------------------------------------------------------------------------ */
#define ORC_DEFINE_BLELLOCH(NAME, T)                                  \
    void NAME(T* arr, int n) {                                        \
        int offset = 1;                                               \
        for (int d = n >> 1; d > 0; d >>= 1) {                        \
            for (int t = 0; t < d; t++) {                             \
                const int ai = offset * (2 * t + 1) - 1;              \
                const int bi = offset * (2 * t + 2) - 1;              \
                arr[bi] += arr[ai];                                   \
            }                                                         \
            offset *= 2;                                              \
        }                                                             \
        arr[n - 1] = 0;                                               \
        for (int d = 1; d < n; d *= 2) {                              \
            offset >>= 1;                                             \
            for (int t = 0; t < d; t++) {                             \
                const int ai = offset * (2 * t + 1) - 1;              \
                const int bi = offset * (2 * t + 2) - 1;              \
                const T tmp = arr[ai];                                \
                arr[ai] = arr[bi];                                    \
                arr[bi] += tmp;                                       \
            }                                                         \
        }                                                             \
    }
ORC_DEFINE_BLELLOCH(orc_blelloch_f32, float)
ORC_DEFINE_BLELLOCH(orc_blelloch_i32, int32_t)
ORC_DEFINE_BLELLOCH(orc_blelloch_i64, int64_t)

/* ------------------------------------------------------------------------------------ */
/* ComputeObjectLUT: warp_prefix_sum (:236-273), ComputePrefixSumWarp2 (:275-296),        */
/* kernel (:959-978).  One warp = 32 lanes; the shuffles are emulated on a lane array.    */
/* ------------------------------------------------------------------------------------ */
void orc_object_lut_column(const float* disp_col, const float* obj_cost_lut, float* lut,
                           const is_stixel_params* p, int n_power2) {
    for (int fn = 0; fn < p->max_dis; fn++) {
        float* arr = &lut[(size_t)fn * (p->rows_power2 + 1)];
        float add = 0.0f;
        arr[0] = 0.0f; /* :283-285 */
        for (int i = 0; i < n_power2; i += WARP) { /* :292-295 */
            float cost[WARP], n[WARP];
            for (int lane = 0; lane < WARP; lane++) {
                int dis = 0; /* :244-247 */
                if (i + lane < p->rows) dis = (int)disp_col[i + lane];
                cost[lane] = obj_cost_lut[fn * p->max_dis + dis];
                if (lane == 0) cost[lane] += add; /* :249-251 */
            }
            for (int j = 1; j < WARP; j *= 2) { /* :255-263 */
                for (int lane = 0; lane < WARP; lane++)
                    n[lane] = (lane >= j) ? cost[lane - j] : cost[lane]; /* __shfl_up */
                for (int lane = 0; lane < WARP; lane++)
                    if (lane >= j) cost[lane] += n[lane];
            }
            /* :266.  The reference stores all 32 lanes, which overruns its (rows_power2+1)-float
             * row when rows < 31; such shapes are outside its domain, we just do not overrun. */
            for (int lane = 0; lane < WARP; lane++)
                if (i + lane + 1 <= p->rows_power2) arr[i + lane + 1] = cost[lane];
            add = cost[WARP - 1]; /* :268-272 */
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* Device cost helpers, StixelsKernels.cu:31-234 and Cityscapes.h:28-123                 */
/* ------------------------------------------------------------------------------------ */
static inline float FastLogDev(float v) { return is_logf(v); }                  /* :31-33 */
static inline float NegFastLogDiv(float v, float v2) {                           /* :35-38 */
    return -FastLogDev(v) + FastLogDev(v2);
}
static inline float GetPriorCost(int vB, int rows) {                             /* :40-42 */
    return NegFastLogDiv(1.0f, (float)(rows - vB));
}

static inline float ComputeMean(int vB, int vT, const float* d_sum, const float* d_valid,
                                float invalid_disparity) { /* :47-60 */
    float mean = 0;
    if (invalid_disparity >= 0) {
        const float valid_dif = d_valid[vT + 1] - d_valid[vB];
        mean = (valid_dif == 0) ? 0 : (d_sum[vT + 1] - d_sum[vB]) / valid_dif;
    } else {
        mean = (d_sum[vT + 1] - d_sum[vB]) / (vT + 1 - vB);
    }
    return mean;
}

static inline int DownsampledSum(const int32_t* column_start, int vB, int vT) { /* Cityscapes.h:28-42 */
    const int vTmod = vT % DSF, vTdiv = vT / DSF;
    const int vBmod = vB % DSF, vBdiv = vB / DSF;
    return (column_start[vTdiv] - column_start[vBdiv]) * DSF +
           (column_start[vTdiv + 1] - column_start[vTdiv]) * (vTmod + 1) -
           (column_start[vBdiv + 1] - column_start[vBdiv]) * vBmod;
}

static inline float ComputeNonInstanceOffsetCost(int vB, int vT, const int32_t* offx_ps,
                                                 const int32_t* offy_ps) { /* :62-70 */
    float cost = DownsampledSum(offx_ps, vB, vT) + DownsampledSum(offy_ps, vB, vT);
    return cost;
}

static inline float ComputeInstanceOffsetCost(int vB, int vT, const int64_t* mx,
                                              const int64_t* my, const int64_t* mx2,
                                              const int64_t* my2) { /* :72-86 */
    const float meanx = mx[vT + 1] - mx[vB];
    const float meany = my[vT + 1] - my[vB];
    const float meanx2 = mx2[vT + 1] - mx2[vB];
    const float meany2 = my2[vT + 1] - my2[vB];
    const float height = vT + 1.0 - vB;
    float cost = meanx2 - meanx * meanx / height + meany2 - meany * meany / height;
    return cost;
}

static inline float GetPriorCostSkyFromObject(float previous_mean, float epsilon,
                                              float prior_cost) { /* :88-96 */
    float cost = is_logf(2.0f) + prior_cost;
    if (previous_mean < epsilon) cost = ORC_INF;
    return cost;
}
static inline float GetPriorCostSkyFromGround(int vB, const float* ground_function,
                                              float prior_cost) { /* :98-106 */
    const float prev_gf = ground_function[vB - 1];
    return (prev_gf < 1.0f) ? prior_cost : ORC_INF;
}
static inline float GetPriorCostObjectFromGround(int vB, float fn, float max_disf,
                                                 const float* ground_function, float prior_cost,
                                                 float epsilon, float pgrav, float pblg) { /* :120-144 */
    float cost = -is_logf(0.7f) + prior_cost;
    float fn_previous = ground_function[vB - 1];
    if (fn_previous < 0.0f) fn_previous = 0.0f;
    if (fn > (fn_previous + epsilon)) {
        cost += NegFastLogDiv(pgrav, max_disf - fn_previous - epsilon);
    } else if (fn < (fn_previous - epsilon)) {
        const float pmean_sub = fn_previous - epsilon;
        cost += NegFastLogDiv(pblg, pmean_sub);
    } else {
        cost += NegFastLogDiv(1.0f - pgrav - pblg, 2.0f * epsilon);
    }
    return cost;
}
static inline float GetPriorCostObjectFromObject(int vB, float fn, float previous_mean,
                                                 const float* object_disparity_range, int vhor,
                                                 float max_disf, float pord, float prior_cost) { /* :146-171 */
    const int previous_vT = vB - 1;
    float cost = (previous_vT < vhor) ? -is_logf(0.7f) : is_logf(2.0f);
    cost += prior_cost;
    float dif_dis = object_disparity_range[(int)previous_mean];
    if (dif_dis < 0.0f) dif_dis = 0.0f;
    if (fn > (previous_mean + dif_dis)) {
        cost += NegFastLogDiv(pord, max_disf - previous_mean - dif_dis);
    } else if (fn < (previous_mean - dif_dis)) {
        const float pmean_sub = previous_mean - dif_dis;
        cost += NegFastLogDiv(1.0f - pord, pmean_sub);
    } else {
        cost = ORC_INF;
    }
    return cost;
}
static inline float GetPriorCostObjectFromSky(float fn, float max_disf, float prior_cost,
                                              float epsilon) { /* :173-183 */
    float cost = ORC_INF;
    if (fn > epsilon) cost = NegFastLogDiv(1.0f, max_disf - epsilon) + prior_cost;
    return cost;
}
static inline float GetPriorCostGround(float prior_cost) { return -is_logf(0.3f) + prior_cost; } /* :185-187 */
static inline float GetPriorCostObjectFirst(int below_vhor_vT, float rows_log, float max_dis_log) { /* :189-194 */
    const float pvt = below_vhor_vT ? is_logf(2.0f) : 0.0f;
    return rows_log + pvt + max_dis_log;
}
static inline float GetPriorCostGroundFirst(float rows_log) { return is_logf(2.0f) + rows_log; } /* :196-199 */

static inline float GetDataCostSky(float d, const is_stixel_params* p) { /* :201-215 */
    float data_cost = p->pnexists_given_sky_log;
    if (d != p->invalid_disparity) {
        const float pgaussian = p->normalization_sky + d * d * p->inv_sigma2_sky;
        const float p_data = fminf(p->puniform_sky, pgaussian);
        data_cost = p_data + p->nopnexists_given_sky_log;
    }
    return data_cost;
}
static inline float GetDataCostGround(float fn, int v, float d, const is_stixel_params* p,
                                      const float* normalization_ground,
                                      const float* inv_sigma2_ground) { /* :217-234 */
    float data_cost = p->pnexists_given_ground_log;
    if (d != p->invalid_disparity) {
        const float model_diff = (d - fn);
        const float pgaussian =
            normalization_ground[v] + model_diff * model_diff * inv_sigma2_ground[v];
        const float p_data = fminf(p->puniform, pgaussian);
        data_cost = p_data + p->nopnexists_given_ground_log;
    }
    return data_cost;
}

/* Cityscapes.h:44-123 */
static inline float GetGroundSegmentationCost(const int32_t* cs, int vB, int vT, int rp2) {
    const float cost_road = DownsampledSum(&cs[0], vB, vT);
    const float cost_sidewalk = DownsampledSum(&cs[rp2], vB, vT);
    return fminf(cost_road, cost_sidewalk);
}
static inline int GetGroundSegmentationClass(const int32_t* cs, int vB, int vT, int rp2) {
    const float cost_road = DownsampledSum(&cs[0], vB, vT);
    const float cost_sidewalk = DownsampledSum(&cs[rp2], vB, vT);
    return (cost_road < cost_sidewalk) ? 0 : 1;
}
static inline float GetObjectSegmentationCost(const int32_t* cs, int vB, int vT, int rp2,
                                              float instance_cost, float non_instance_cost,
                                              int* min_class_out) {
    float min_cost_segmentation = ORC_INF;
    int min_class = 2;
    for (int c = 2; c < 19; c++) {
        float cost_segmentation = 0.0f;
        if (c < 10)
            cost_segmentation += non_instance_cost;
        else if (c == 10)
            continue;
        else
            cost_segmentation += instance_cost;
        cost_segmentation += DownsampledSum(&cs[c * rp2], vB, vT);
        if (min_cost_segmentation > cost_segmentation) {
            min_cost_segmentation = cost_segmentation;
            min_class = c;
        }
    }
    if (min_class_out) *min_class_out = min_class;
    return min_cost_segmentation;
}
static inline float GetSkySegmentationCost(const int32_t* cs, int vB, int vT, int rp2) {
    return DownsampledSum(&cs[10 * rp2], vB, vT);
}

/* ------------------------------------------------------------------------------------ */
/* StixelsKernel<PAIRWISE>, StixelsKernels.cu:298-957, one column (= one thread block)    */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    float *sky_lut, *ground_lut, *ground_function, *cost_table;
    int16_t* index_table;
    uint8_t* index_written;
    float *disparity_prefixsum, *valid_disparity;
    int64_t *mx_ps, *my_ps, *mx2_ps, *my2_ps;
    int32_t* seg;      /* working copy of this column's [channels][P2S] tensor */
    float* object_lut; /* [max_dis][P2+1] */
} colws;

static colws* ws_alloc(const is_stixel_params* p) {
    colws* w = (colws*)calloc(1, sizeof(colws));
    const size_t n = (size_t)p->rows_power2;
    w->sky_lut = (float*)calloc(n, sizeof(float));
    w->ground_lut = (float*)calloc(n, sizeof(float));
    w->ground_function = (float*)calloc(n, sizeof(float));
    w->cost_table = (float*)calloc(n * 3, sizeof(float));
    w->index_table = (int16_t*)calloc(n * 3, sizeof(int16_t));
    w->index_written = (uint8_t*)calloc(n * 3, 1);
    w->disparity_prefixsum = (float*)calloc(n, sizeof(float));
    w->valid_disparity = (float*)calloc(n, sizeof(float));
    w->mx_ps = (int64_t*)calloc(n, sizeof(int64_t));
    w->my_ps = (int64_t*)calloc(n, sizeof(int64_t));
    w->mx2_ps = (int64_t*)calloc(n, sizeof(int64_t));
    w->my2_ps = (int64_t*)calloc(n, sizeof(int64_t));
    w->seg = (int32_t*)calloc((size_t)p->segmentation_channels * p->rows_power2_segmentation,
                              sizeof(int32_t));
    w->object_lut = (float*)calloc((size_t)p->max_dis * (n + 1), sizeof(float));
    return w;
}
static void ws_free(colws* w) {
    free(w->sky_lut); free(w->ground_lut); free(w->ground_function); free(w->cost_table);
    free(w->index_table); free(w->index_written); free(w->disparity_prefixsum);
    free(w->valid_disparity); free(w->mx_ps); free(w->my_ps); free(w->mx2_ps); free(w->my2_ps);
    free(w->seg); free(w->object_lut); free(w);
}

typedef struct { /* per-column instance candidates, emitted in section order */
    int n;
    int cls[IS_MAX_STIXELS_PER_COLUMN];
    int sec[IS_MAX_STIXELS_PER_COLUMN];
    float mx[IS_MAX_STIXELS_PER_COLUMN], my[IS_MAX_STIXELS_PER_COLUMN];
    uint8_t core[IS_MAX_STIXELS_PER_COLUMN];
} col_instances;

static void set_index(colws* w, int idx, int value) {
    w->index_table[idx] = (int16_t)value;
    w->index_written[idx] = 1;
}

static void stixels_column(const is_stixel_params* pp, int pairwise, int col,
                           const float* d_disparity_col, const int32_t* seg_in,
                           const float* d_ground_function, const float* d_normalization_ground,
                           const float* d_inv_sigma2_ground,
                           const float* object_disparity_range, colws* w, is_section* d_stixels,
                           col_instances* inst) {
    const is_stixel_params params = *pp;
    const int rows = params.rows;
    const int P2 = params.rows_power2;
    const int P2S = params.rows_power2_segmentation;
    const int CH = params.segmentation_channels;
    const float prior_weight = params.prior_weight;
    const float disparity_weight = params.disparity_weight;
    const float segmentation_weight = params.segmentation_weight;

    /* private copy: the reference mutates its input in place */
    memcpy(w->seg, seg_in, sizeof(int32_t) * (size_t)CH * P2S);
    int32_t* d_segmentation = w->seg; /* column base */
    /* smem arrays are uninitialised beyond `rows` in the reference (Q4): zero them */
    memset(w->sky_lut, 0, sizeof(float) * P2);
    memset(w->ground_lut, 0, sizeof(float) * P2);
    memset(w->disparity_prefixsum, 0, sizeof(float) * P2);
    memset(w->valid_disparity, 0, sizeof(float) * P2);
    memset(w->mx_ps, 0, sizeof(int64_t) * P2);
    memset(w->my_ps, 0, sizeof(int64_t) * P2);
    memset(w->mx2_ps, 0, sizeof(int64_t) * P2);
    memset(w->my2_ps, 0, sizeof(int64_t) * P2);
    memset(w->index_written, 0, (size_t)P2 * 3);
    memset(w->index_table, 0, sizeof(int16_t) * (size_t)P2 * 3);

    /* offsets of the two instance-offset channels inside the column block, :393-398, :418-420 */
    const int off_base = params.segmentation_classes * P2S; /* UV_OFFSET = 0 */
    int32_t* instance_offsetsy_ps = &d_segmentation[off_base];
    int32_t* instance_offsetsx_ps = &d_segmentation[off_base + P2S];

    /* ---- load phase, :371-446.  First all reads of raw offsets (Q3 lockstep), ... */
    for (int row = 0; row < rows; row++) {
        const float d = d_disparity_col[row];
        w->cost_table[row] = ORC_INF; /* :374-376 */
        w->cost_table[rows + row] = ORC_INF;
        w->cost_table[2 * rows + row] = ORC_INF;

        if (params.invalid_disparity >= 0) { /* :382-389 */
            const int va = d != params.invalid_disparity;
            w->valid_disparity[row] = (float)va;
            w->disparity_prefixsum[row] = ((float)va) * d;
        } else {
            w->disparity_prefixsum[row] = d;
        }

        const int row_index = off_base + row / DSF;
        /* :401-409 (double arithmetic, truncation toward zero on the int64 store) */
        w->mx_ps[row] = (int64_t)((params.column_step * col + 0.5 * (params.column_step - 1.0)) +
                                  d_segmentation[row_index + P2S] + 0.5);
        w->my_ps[row] = (int64_t)(row - d_segmentation[row_index] + 0.5);
        w->mx2_ps[row] = w->mx_ps[row] * w->mx_ps[row];
        w->my2_ps[row] = w->my_ps[row] * w->my_ps[row];

        /* :424-446 */
        w->sky_lut[row] = (row < params.vhor) ? 0 : GetDataCostSky(d, &params);
        w->ground_function[row] = d_ground_function[row];
        const float gf = w->ground_function[row];
        w->ground_lut[row] = (row >= params.vhor)
                                 ? ORC_INF
                                 : GetDataCostGround(gf, row, d, &params, d_normalization_ground,
                                                     d_inv_sigma2_ground);
    }
    /* ... then the in-place squaring of the offset channels, :411-416 */
    for (int row = 0; row < rows; row++) {
        if (row % DSF == 0) {
            const int row_index = off_base + row / DSF;
            d_segmentation[row_index + P2S] *= d_segmentation[row_index + P2S];
            d_segmentation[row_index] *= d_segmentation[row_index];
        }
    }

    /* ---- prefix sums, :452-469 */
    if (params.invalid_disparity >= 0) orc_blelloch_f32(w->valid_disparity, P2);
    orc_blelloch_f32(w->disparity_prefixsum, P2);
    orc_blelloch_i64(w->mx_ps, P2);
    orc_blelloch_i64(w->my_ps, P2);
    orc_blelloch_i64(w->mx2_ps, P2);
    orc_blelloch_i64(w->my2_ps, P2);
    orc_blelloch_f32(w->ground_lut, P2);
    orc_blelloch_f32(w->sky_lut, P2);
    for (int c = 0; c < params.segmentation_classes + 2; c++)
        orc_blelloch_i32(&d_segmentation[c * P2S], P2S);

    const float max_disf = (float)params.max_dis; /* :472 */
    float* cost_table = w->cost_table;
    const float* ground_lut = w->ground_lut;
    const float* sky_lut = w->sky_lut;
    const float* d_object_lut = w->object_lut; /* obj_data_idx = 0 for the private column LUT */
    const int lut_stride = P2 + 1;

    /* ---- first segment, vB = 0, :481-594 (all threads vT in parallel) */
    for (int vT = 0; vT < rows; vT++) {
        const int vB = 0;
        const float inverse_height = 1. / (vT + 1 - vB);
        const float instance_cost =
            params.instance_weight *
            ComputeInstanceOffsetCost(vB, vT, w->mx_ps, w->my_ps, w->mx2_ps, w->my2_ps);
        const float non_instance_cost =
            params.instance_weight *
            ComputeNonInstanceOffsetCost(vB, vT, instance_offsetsx_ps, instance_offsetsy_ps);
        const float cost_ground_segmentation =
            GetGroundSegmentationCost(d_segmentation, vB, vT, P2S) + non_instance_cost;
        const float cost_object_segmentation = GetObjectSegmentationCost(
            d_segmentation, vB, vT, P2S, instance_cost, non_instance_cost, NULL);

        float obj_fn = ComputeMean(vB, vT, w->disparity_prefixsum, w->valid_disparity,
                                   params.invalid_disparity);
        if (obj_fn < 0) obj_fn = 0;
        const int obj_fni = (int)floorf(obj_fn);

        const float cost_ground_data = ground_lut[vT + 1] - ground_lut[vB];
        const float cost_object_data = d_object_lut[obj_fni * lut_stride + vT + 1] -
                                       d_object_lut[obj_fni * lut_stride + vB];

        const int index_pground = vT * 3 + IS_GROUND;
        const int index_pobject = vT * 3 + IS_OBJECT;
        const int below_vhor_vT = vT <= params.vhor;

        if (below_vhor_vT) { /* :545-566 */
            const float curr_cost_ground = cost_table[index_pground];
            float cost_ground;
            if (pairwise) {
                const float cost_ground_prior = GetPriorCostGroundFirst(params.rows_log);
                cost_ground = disparity_weight * cost_ground_data +
                              prior_weight * cost_ground_prior +
                              segmentation_weight * cost_ground_segmentation;
            } else {
                cost_ground = disparity_weight * cost_ground_data +
                              prior_weight * inverse_height +
                              segmentation_weight * cost_ground_segmentation;
            }
            if (cost_ground < curr_cost_ground) {
                cost_table[index_pground] = cost_ground;
                set_index(w, index_pground, IS_GROUND);
            }
        }

        const float curr_cost_object = cost_table[index_pobject]; /* :569-592 */
        float cost_object;
        if (pairwise) {
            const float cost_object_prior =
                GetPriorCostObjectFirst(below_vhor_vT, params.rows_log, params.max_dis_log);
            cost_object = disparity_weight * cost_object_data +
                          prior_weight * cost_object_prior +
                          segmentation_weight * cost_object_segmentation;
        } else {
            cost_object = disparity_weight * cost_object_data + prior_weight * inverse_height +
                          segmentation_weight * cost_object_segmentation;
        }
        if (cost_object < curr_cost_object) cost_table[index_pobject] = cost_object;
        set_index(w, index_pobject, IS_OBJECT);
    }

    /* ---- vB >= 1, :600-839.  One outer iteration = one __syncthreads() region. */
    for (int vB = 1; vB < rows; vB++) {
        /* values that are uniform over the block in this region */
        const int previous_vT = vB - 1;
        const int below_vhor_vTprev = previous_vT < params.vhor;
        /* the region reads cost_table[previous_vT*3+*] and writes cost_table[vT*3+*], vT>=vB:
         * disjoint, so thread order inside the region is irrelevant */
        for (int vT = vB; vT < rows; vT++) {
            const float inverse_height = 1. / (vT + 1 - vB);
            const float instance_cost =
                params.instance_weight *
                ComputeInstanceOffsetCost(vB, vT, w->mx_ps, w->my_ps, w->mx2_ps, w->my2_ps);
            const float non_instance_cost =
                params.instance_weight *
                ComputeNonInstanceOffsetCost(vB, vT, instance_offsetsx_ps, instance_offsetsy_ps);

            const float cost_ground_segmentation =
                GetGroundSegmentationCost(d_segmentation, vB, vT, P2S) + non_instance_cost;
            const float cost_object_segmentation = GetObjectSegmentationCost(
                d_segmentation, vB, vT, P2S, instance_cost, non_instance_cost, NULL);
            const float cost_sky_segmentation =
                GetSkySegmentationCost(d_segmentation, vB, vT, P2S) + non_instance_cost;

            float obj_fn = ComputeMean(vB, vT, w->disparity_prefixsum, w->valid_disparity,
                                       params.invalid_disparity);
            if (obj_fn < 0) obj_fn = 0;
            const int obj_fni = (int)floorf(obj_fn);

            const float cost_object_data = d_object_lut[obj_fni * lut_stride + vT + 1] -
                                           d_object_lut[obj_fni * lut_stride + vB];
            float prior_cost = 0;
            if (pairwise) prior_cost = GetPriorCost(vB, params.rows);

            float previous_mean = 0;
            if (pairwise) { /* :675-685 */
                const int previous_object_vB = w->index_table[previous_vT * 3 + IS_OBJECT] / 3;
                previous_mean = ComputeMean(previous_object_vB, previous_vT,
                                            w->disparity_prefixsum, w->valid_disparity,
                                            params.invalid_disparity);
                if (previous_mean < 0) previous_mean = 0;
            }

            if (below_vhor_vTprev) { /* ground, :687-728 */
                const float cost_ground_data = ground_lut[vT + 1] - ground_lut[vB];
                const int index_pground = vT * 3 + IS_GROUND;
                const float curr_cost_ground = cost_table[index_pground];
                float cost_ground_prior1 = cost_table[previous_vT * 3 + IS_GROUND];
                float cost_ground_prior2 = cost_table[previous_vT * 3 + IS_OBJECT];
                float cost_ground;
                if (pairwise) {
                    const float prev_cost = GetPriorCostGround(prior_cost);
                    cost_ground_prior1 += prior_weight * prev_cost;
                    cost_ground_prior2 += prior_weight * prev_cost;
                    const float cost_ground_minprior =
                        fminf(cost_ground_prior1, cost_ground_prior2);
                    cost_ground = disparity_weight * cost_ground_data +
                                  prior_weight * cost_ground_minprior +
                                  segmentation_weight * cost_ground_segmentation;
                } else {
                    cost_ground = disparity_weight * cost_ground_data +
                                  prior_weight * inverse_height +
                                  segmentation_weight * cost_ground_segmentation;
                }
                if (cost_ground < curr_cost_ground) {
                    cost_table[index_pground] = cost_ground;
                    int min_prev = IS_OBJECT;
                    if (cost_ground_prior1 < cost_ground_prior2) min_prev = IS_GROUND;
                    set_index(w, index_pground, vB * 3 + min_prev);
                }
            } else { /* sky, :729-775 */
                const float cost_sky_data = sky_lut[vT + 1] - sky_lut[vB];
                const int index_psky = vT * 3 + IS_SKY;
                const float curr_cost_sky = cost_table[index_psky];
                float cost_sky_prior1 = cost_table[previous_vT * 3 + IS_GROUND];
                float cost_sky_prior2 = cost_table[previous_vT * 3 + IS_OBJECT];
                float cost_sky;
                if (pairwise) {
                    cost_sky_prior1 +=
                        prior_weight *
                        GetPriorCostSkyFromGround(vB, w->ground_function, prior_cost);
                    cost_sky_prior2 +=
                        prior_weight *
                        GetPriorCostSkyFromObject(previous_mean, params.epsilon, prior_cost);
                    const float cost_sky_minprior = fminf(cost_sky_prior1, cost_sky_prior2);
                    cost_sky = disparity_weight * cost_sky_data +
                               prior_weight * cost_sky_minprior +
                               segmentation_weight * cost_sky_segmentation;
                } else {
                    cost_sky = disparity_weight * cost_sky_data + prior_weight * inverse_height +
                               segmentation_weight * cost_sky_segmentation;
                }
                if (cost_sky < curr_cost_sky) {
                    cost_table[index_psky] = cost_sky;
                    int min_prev = IS_OBJECT;
                    if (cost_sky_prior1 < cost_sky_prior2) min_prev = IS_GROUND;
                    set_index(w, index_psky, vB * 3 + min_prev);
                }
            }

            /* object, :777-837 */
            const int index_pobject = vT * 3 + IS_OBJECT;
            const float curr_cost_object = cost_table[index_pobject];
            float cost_object;
            float cost_object_prior1 = cost_table[previous_vT * 3 + IS_GROUND];
            float cost_object_prior2 = cost_table[previous_vT * 3 + IS_OBJECT];
            float cost_object_prior3 = cost_table[previous_vT * 3 + IS_SKY];
            if (pairwise) {
                cost_object_prior1 +=
                    prior_weight * GetPriorCostObjectFromGround(vB, obj_fn, max_disf,
                                                                w->ground_function, prior_cost,
                                                                params.epsilon, params.pgrav,
                                                                params.pblg);
                cost_object_prior2 +=
                    prior_weight * GetPriorCostObjectFromObject(vB, obj_fn, previous_mean,
                                                                object_disparity_range,
                                                                params.vhor, max_disf,
                                                                params.pord, prior_cost);
                cost_object_prior3 +=
                    prior_weight *
                    GetPriorCostObjectFromSky(obj_fn, max_disf, prior_cost, params.epsilon);
                const float cost_object_minprior =
                    fminf(fminf(cost_object_prior1, cost_object_prior2), cost_object_prior3);
                cost_object = disparity_weight * cost_object_data +
                              prior_weight * cost_object_minprior +
                              segmentation_weight * cost_object_segmentation;
            } else {
                cost_object = disparity_weight * cost_object_data +
                              prior_weight * inverse_height +
                              segmentation_weight * cost_object_segmentation;
            }
            if (cost_object < curr_cost_object) {
                cost_table[index_pobject] = cost_object;
                int min_prev = IS_OBJECT;
                if (cost_object_prior1 < cost_object_prior2) min_prev = IS_GROUND;
                if (cost_object_prior3 < fminf(cost_object_prior1, cost_object_prior2))
                    min_prev = IS_SKY;
                set_index(w, index_pobject, vB * 3 + min_prev);
            }
        }
    }

    /* ---- backtracing by thread 0, :843-955 */
    inst->n = 0;
    {
        int vT = rows - 1;
        const float last_ground = cost_table[vT * 3 + IS_GROUND];
        const float last_object = cost_table[vT * 3 + IS_OBJECT];
        const float last_sky = cost_table[vT * 3 + IS_SKY];
        int type = IS_OBJECT;
        if (last_ground < last_object) type = IS_GROUND;
        if (last_sky < fminf(last_ground, last_object)) type = IS_SKY;
        int min_idx = vT * 3 + type;
        int prev_vT;
        int i = 0;
        do {
            prev_vT = (w->index_table[min_idx] / 3) - 1;
            is_section sec;
            sec.vT = vT;
            sec.type = type;
            sec.vB = prev_vT + 1;
            sec.disparity = (float)ComputeMean(sec.vB, sec.vT, w->disparity_prefixsum,
                                               w->valid_disparity, params.invalid_disparity);
            sec.cost = fminf(cost_table[sec.vT * 3 + type], 1e4);
            sec.instance_meanx =
                (float)(w->mx_ps[sec.vT + 1] - w->mx_ps[sec.vB]) / (sec.vT + 1 - sec.vB);
            sec.instance_meany =
                (float)(w->my_ps[sec.vT + 1] - w->my_ps[sec.vB]) / (sec.vT + 1 - sec.vB);

            if (sec.type == IS_GROUND) {
                sec.semantic_class = GetGroundSegmentationClass(d_segmentation, sec.vB, sec.vT, P2S);
            } else if (sec.type == IS_SKY || sec.disparity < 1.0) {
                sec.type = IS_SKY;
                sec.semantic_class = 10; /* GetSkySegmentationClass, Cityscapes.h:119-122 */
            } else {
                const float instance_cost =
                    params.instance_weight * ComputeInstanceOffsetCost(sec.vB, sec.vT, w->mx_ps,
                                                                       w->my_ps, w->mx2_ps,
                                                                       w->my2_ps);
                const float non_instance_cost =
                    params.instance_weight *
                    ComputeNonInstanceOffsetCost(sec.vB, sec.vT, instance_offsetsx_ps,
                                                 instance_offsetsy_ps);
                int cls;
                (void)GetObjectSegmentationCost(d_segmentation, sec.vB, sec.vT, P2S,
                                                instance_cost, non_instance_cost, &cls);
                sec.semantic_class = cls; /* GetObjectSegmentationClass, Cityscapes.h:85-111 */
                if (sec.semantic_class >= IS_FIRST_INSTANCE_CLASS) { /* :926-942 */
                    const int k = inst->n++;
                    inst->cls[k] = sec.semantic_class - IS_FIRST_INSTANCE_CLASS;
                    inst->sec[k] = i;
                    inst->mx[k] = sec.instance_meanx;
                    inst->my[k] = sec.instance_meany;
                    inst->core[k] = (sec.vT + 1 - sec.vB) >= params.clustering_size_filter;
                }
            }
            d_stixels[i] = sec;

            type = w->index_table[min_idx] % 3;
            vT = prev_vT;
            min_idx = prev_vT * 3 + type;
            i++;
            /* the reference asserts i < max_sections (:950); we stop instead of overflowing */
        } while (prev_vT != -1 && i < params.max_sections - 1);
        is_section term;
        memset(&term, 0, sizeof(term));
        term.type = -1;
        d_stixels[i] = term;
    }
}

int orc_compute(const is_stixel_params* p, const float* obj_cost_lut,
                const float* obj_disparity_range, const float* disp_joined, const int32_t* seg,
                const float* ground_function, const float* normalization_ground,
                const float* inv_sigma2_ground, int pairwise, int col_begin, int col_end,
                int nthreads, is_section* sections, float* cost_table_out,
                int32_t* index_table_out, float* inst_centerofmass, int32_t* inst_indices,
                uint8_t* inst_core, int32_t* inst_per_class) {
    if (p->column_step != DSF) return -1; /* assert, StixelsKernels.cu:318 */
    const int rows = p->rows;
    const int realcols = p->cols;
    const int S = p->max_sections;
    const int CH = p->segmentation_channels;
    const int P2S = p->rows_power2_segmentation;
    /* n_power2 argument of ComputeObjectLUT, Stixels.cu:537 */
    const int n_power2 = (int)powf(2, ceilf(log2f(rows)));
    col_instances* all_inst = (col_instances*)calloc((size_t)realcols, sizeof(col_instances));
    if (nthreads < 1) nthreads = 1;

#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
    {
        colws* w = ws_alloc(p);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int col = col_begin; col < col_end; col++) {
            const float* dcol = &disp_joined[(size_t)col * rows];
            orc_object_lut_column(dcol, obj_cost_lut, w->object_lut, p, n_power2);
            stixels_column(p, pairwise, col, dcol, &seg[(size_t)col * CH * P2S], ground_function,
                           normalization_ground, inv_sigma2_ground, obj_disparity_range, w,
                           &sections[(size_t)col * S], &all_inst[col]);
            if (cost_table_out)
                memcpy(&cost_table_out[(size_t)col * rows * 3], w->cost_table,
                       sizeof(float) * rows * 3);
            if (index_table_out)
                for (int k = 0; k < rows * 3; k++)
                    index_table_out[(size_t)col * rows * 3 + k] =
                        w->index_written[k] ? (int32_t)w->index_table[k] : -1;
        }
        ws_free(w);
    }

    /* canonical (column, section) order; layout StixelsKernels.cu:931-941 */
    int counts[IS_INSTANCE_CLASSES] = {0};
    for (int col = col_begin; col < col_end; col++) {
        const col_instances* ci = &all_inst[col];
        for (int k = 0; k < ci->n; k++) {
            const int cls = ci->cls[k];
            const int class_offset = cls * realcols * S;
            const int idx = counts[cls]++;
            if (inst_centerofmass) {
                inst_centerofmass[(class_offset + idx) * 2] = ci->mx[k];
                inst_centerofmass[(class_offset + idx) * 2 + 1] = ci->my[k];
            }
            if (inst_indices) {
                inst_indices[(class_offset + idx) * 2] = col;
                inst_indices[(class_offset + idx) * 2 + 1] = ci->sec[k];
            }
            if (inst_core) inst_core[class_offset + idx] = ci->core[k];
        }
    }
    if (inst_per_class)
        for (int c = 0; c < IS_INSTANCE_CLASSES; c++) inst_per_class[c] = counts[c];
    free(all_inst);
    return 0;
}
