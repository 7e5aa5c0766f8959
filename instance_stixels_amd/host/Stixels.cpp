/*
 * Stixels.cpp -- host class of the MI355X-native stixel library (plain C++, no HIP headers).
 *
 * Mirrors the behaviour of the reference's host driver
 * (/root/reference/InstanceStixels/src/Stixels.cu:33-926) on top of the C ABI in
 * include/instance_stixels_core.h: same setters, same frame-independent precompute
 * (Initialize), same per-frame ground model (PrecomputeGround), same Compute() call sequence
 * and error conventions (std::invalid_argument for unset configuration, message + exit(1) for
 * device runtime failures).  Citations `Stixels.cu:N` refer to that file.
 */
#include "InstanceStixels/Stixels.hpp"

#include <algorithm>
#include <cstring>
#include <fstream>
#include <iostream>
#include <stdexcept>

#include "DeviceGuard.h"

Stixels::Stixels() {}
Stixels::~Stixels() {} /* like the reference, buffers are released by Finish(), Stixels.cu:36-37 */

/* ---------------------------------------------------------------- configuration setters */

void Stixels::SetConfig(const StixelConfig& config) { /* Stixels.cu:292-338 */
    if (config.rows == -1 || config.cols == -1)
        throw std::invalid_argument("Number of rows or columns are not set.");
    if (config.max_dis == -1) throw std::invalid_argument("Maximum disparity value is not set.");
    if (config.eps == -1 || config.min_pts == -1 || config.size_filter == -1)
        throw std::invalid_argument("Clustering parameters are not set.");
    if (config.prior_weight == -1 || config.segmentation_weight == -1 ||
        config.instance_weight == -1 || config.disparity_weight == -1)
        throw std::invalid_argument("Energy term weights are not set.");
    if (config.column_step == -1) throw std::invalid_argument("Stixel width is not set.");
    if (config.focal == -1 || config.baseline == -1)
        throw std::invalid_argument("Camera parameters are not set.");

    SetDisparityParameters(config.rows, config.cols, config.max_dis, config.invalid_disparity,
                           config.sigma_disparity_object, config.sigma_disparity_ground,
                           config.sigma_sky);
    SetSegmentationParameters(config.n_semantic_classes, config.n_offset_channels);
    SetClusteringParameters(config.eps, config.min_pts, config.size_filter);
    SetWeightParameters(config.prior_weight, config.disparity_weight, config.segmentation_weight,
                        config.instance_weight);
    SetProbabilities(config.pout, config.pout_sky, config.pground_given_nexist,
                     config.pobject_given_nexist, config.psky_given_nexist, config.pnexist_dis,
                     config.pground, config.pobject, config.psky, config.pord, config.pgrav,
                     config.pblg);
    SetModelParameters(config.column_step, config.median_join, config.epsilon,
                       config.range_objects_z, config.width_margin);
    SetCameraParameters(config.focal, config.baseline, config.sigma_camera_tilt,
                        config.sigma_camera_height, config.camera_center_x,
                        config.camera_center_y);
}

void Stixels::SetProbabilities(float pout, float pout_sky, float pground_given_nexist,
                               float pobject_given_nexist, float psky_given_nexist,
                               float pnexist_dis, float pground, float pobject, float psky,
                               float pord, float pgrav, float pblg) { /* Stixels.cu:361-373 */
    m_pout = pout;
    m_pout_sky = pout_sky;
    m_pnexists_given_ground = (pground_given_nexist * pnexist_dis) / pground;
    m_pnexists_given_object = (pobject_given_nexist * pnexist_dis) / pobject;
    m_pnexists_given_sky = (psky_given_nexist * pnexist_dis) / psky;
    m_pord = pord;
    m_pgrav = pgrav;
    m_pblg = pblg;
}

void Stixels::SetRoadParameters(int vhor, float camera_tilt, float camera_height,
                                float alpha_ground) { /* Stixels.cu:375-381 */
    m_vhor = m_rows - vhor - 1;
    m_camera_tilt = camera_tilt;
    m_camera_height = camera_height;
    m_alpha_ground = alpha_ground;
}

void Stixels::SetCameraParameters(float focal, float baseline, float sigma_camera_tilt,
                                  float sigma_camera_height, float camera_center_x,
                                  float camera_center_y) { /* Stixels.cu:383-393 */
    m_focal = focal;
    m_baseline = baseline;
    m_sigma_camera_tilt = sigma_camera_tilt * (PIFLOAT) / 180.0f; /* degrees -> radians */
    m_sigma_camera_height = sigma_camera_height;
    m_camera_center_x = camera_center_x;
    m_camera_center_y = camera_center_y;
}

void Stixels::SetClusteringParameters(const float eps, const int min_pts,
                                      const int size_filter) { /* Stixels.cu:395-400 */
    m_params.clustering_eps = eps;
    m_params.clustering_min_pts = min_pts;
    m_params.clustering_size_filter = size_filter;
}

void Stixels::SetSegmentationParameters(const int classes,
                                        const int instance_channels) { /* Stixels.cu:402-406 */
    m_segmentation_classes = classes;
    m_segmentation_channels = classes + instance_channels;
}

void Stixels::SetWeightParameters(const float prior_weight, const float disparity_weight,
                                  const float segmentation_weight,
                                  const float instance_weight) { /* Stixels.cu:408-423 */
    m_prior_weight = prior_weight;
    m_disparity_weight = disparity_weight;
    m_segmentation_weight = segmentation_weight;
    /* the instance weight is expressed relative to the segmentation weight */
    m_instance_weight = 0.0;
    if (segmentation_weight > 1e-5) {
        m_instance_weight = instance_weight / segmentation_weight;
        if (instance_weight < 1e-8) m_instance_weight = 0.0;
    }
}

void Stixels::SetDisparityParameters(const int rows, const int cols, const int max_dis,
                                     const float invalid_disparity,
                                     const float sigma_disparity_object,
                                     const float sigma_disparity_ground,
                                     const float sigma_sky) { /* Stixels.cu:425-437 */
    m_rows = rows;
    m_cols = cols;
    m_max_dis = max_dis;
    m_max_disf = (float)m_max_dis;
    m_sigma_disparity_object = sigma_disparity_object;
    m_sigma_disparity_ground = sigma_disparity_ground;
    m_sigma_sky = sigma_sky;
    m_invalid_disparity = invalid_disparity;
}

void Stixels::SetModelParameters(const int column_step, const bool median_join, float epsilon,
                                 float range_objects_z, int width_margin) { /* Stixels.cu:439-446 */
    m_column_step = column_step;
    m_median_join = median_join;
    m_epsilon = epsilon;
    m_range_objects_z = range_objects_z;
    m_width_margin = width_margin;
}

/* ---------------------------------------------------------------- precompute */

float Stixels::FastLog(float v) const { /* Stixels.cu:786-788 */
    return m_log_lut[(int)((v)*LOG_LUT_SIZE + 0.5f)];
}

float Stixels::ComputeObjectDisparityRange(const float previous_mean) const { /* :879-887 */
    float range_disp = 0.0f;
    if (previous_mean != 0) {
        const float pmean_plus_z = (m_baseline * m_focal / previous_mean) + m_range_objects_z;
        range_disp = previous_mean - (m_baseline * m_focal / pmean_plus_z);
    }
    return range_disp;
}

void Stixels::PrecomputeSky() { /* Stixels.cu:856-865 */
    const float sigma = m_sigma_sky;
    const float pout = m_pout_sky;
    const float a_range =
        0.5f * (std::erf(m_max_disf / (sigma * sqrtf(2.0f))) - std::erf(0.0f));
    m_normalization_sky =
        FastLog(a_range) - logf((1.0f - pout) / (sigma * sqrtf(2.0f * PIFLOAT)));
    m_inv_sigma2_sky = 1.0f / (2.0f * sigma * sigma);
}

void Stixels::PrecomputeObject() { /* Stixels.cu:819-840 */
    const float pout = m_pout;
    m_normalization_object.assign(m_max_dis, 0.0f);
    m_inv_sigma2_object.assign(m_max_dis, 0.0f);
    for (int dis = 0; dis < m_max_dis; dis++) {
        const float fn = (float)dis;
        const float sigma_object = fn * fn * m_range_objects_z / (m_focal * m_baseline);
        const float sigma = sqrtf(m_sigma_disparity_object * m_sigma_disparity_object +
                                  sigma_object * sigma_object);
        const float a_range = 0.5f * (std::erf((m_max_disf - fn) / (sigma * sqrtf(2.0f))) -
                                      std::erf((-fn) / (sigma * sqrtf(2.0f))));
        m_normalization_object[dis] =
            FastLog(a_range) - FastLog((1.0f - pout) / (sigma * sqrtf(2.0f * PIFLOAT)));
        m_inv_sigma2_object[dis] = 1.0f / (2.0f * sigma * sigma);
    }
}

float Stixels::GetDataCostObject(const int fn, const int dis) const { /* Stixels.cu:842-854 */
    float data_cost = m_params.pnexists_given_object_log;
    if (dis != (int)m_invalid_disparity) {
        const float model_diff = (float)(dis - fn);
        const float pgaussian =
            m_normalization_object[fn] + model_diff * model_diff * m_inv_sigma2_object[fn];
        const float p_data = fminf(m_puniform, pgaussian);
        data_cost = p_data + m_params.nopnexists_given_object_log;
    }
    return data_cost;
}

void Stixels::PrecomputeGround(int vhor_lib, float camera_tilt, float camera_height,
                               float alpha_ground, GroundModel& out) const { /* :790-817 */
    const float fb = (m_focal * m_baseline) / camera_height;
    const float pout = m_pout;
    out.function.resize(m_rows);
    out.normalization.resize(m_rows);
    out.inv_sigma2.resize(m_rows);
    for (int v = 0; v < m_rows; v++) {
        const float fn = alpha_ground * (float)(vhor_lib - v); /* GroundFunction, :867-877 */
        out.function[v] = fn;
        const float x = camera_tilt + (float)(vhor_lib - v) / m_focal;
        const float sigma2_road =
            fb * fb *
            (m_sigma_camera_height * m_sigma_camera_height * x * x /
                 (camera_height * camera_height) +
             m_sigma_camera_tilt * m_sigma_camera_tilt);
        const float sigma =
            sqrtf(m_sigma_disparity_ground * m_sigma_disparity_ground + sigma2_road);
        const float a_range = 0.5f * (std::erf((m_max_disf - fn) / (sigma * sqrtf(2.0f))) -
                                      std::erf((-fn) / (sigma * sqrtf(2.0f))));
        out.normalization[v] =
            FastLog(a_range) - FastLog((1.0f - pout) / (sigma * sqrtf(2.0f * PIFLOAT)));
        out.inv_sigma2[v] = 1.0f / (2.0f * sigma * sigma);
    }
}

void Stixels::GetGroundModel(std::vector<float>& ground_function,
                             std::vector<float>& normalization_ground,
                             std::vector<float>& inv_sigma2_ground, int& vhor_lib) {
    GroundModel g;
    PrecomputeGround(m_vhor, m_camera_tilt, m_camera_height, m_alpha_ground, g);
    ground_function = g.function;
    normalization_ground = g.normalization;
    inv_sigma2_ground = g.inv_sigma2;
    vhor_lib = m_vhor;
}

/* ---------------------------------------------------------------- Initialize / Finish */

void Stixels::Initialize() { InitializeBatch(1); }

/* Host half of Initialize: every frame-independent table and the kernel parameter block
 * (Stixels.cu:44-47, 79-133, 212-245).  Needs no device. */
void Stixels::PrecomputeHost() {
    m_realcols = (m_cols - m_width_margin) / m_column_step;
    m_max_sections = MAX_STIXELS_PER_COLUMN;
    m_instance_classes = IS_INSTANCE_CLASSES;

    m_instances_per_class.assign(m_instance_classes, 0);

    /* log LUT over [0, 1], Stixels.cu:79-84 */
    m_log_lut.resize(LOG_LUT_SIZE + 1);
    for (int i = 0; i < LOG_LUT_SIZE; i++) {
        const float log_res = (float)i / ((float)LOG_LUT_SIZE);
        m_log_lut[i] = logf(log_res);
    }
    m_log_lut[LOG_LUT_SIZE] = 0.0f;

    /* frequently used values, Stixels.cu:93-102 */
    m_max_dis_log = logf(m_max_disf);
    m_rows_log = logf((float)m_rows);
    m_puniform_sky = m_max_dis_log - logf(m_pout_sky);
    m_puniform = m_max_dis_log - logf(m_pout);
    m_params.pnexists_given_sky_log = -logf(m_pnexists_given_sky);
    m_params.nopnexists_given_sky_log = -logf(1.0f - m_pnexists_given_sky);
    m_params.pnexists_given_ground_log = -logf(m_pnexists_given_ground);
    m_params.nopnexists_given_ground_log = -logf(1.0f - m_pnexists_given_ground);
    m_params.pnexists_given_object_log = -logf(m_pnexists_given_object);
    m_params.nopnexists_given_object_log = -logf(1.0f - m_pnexists_given_object);

    m_object_disparity_range.resize(m_max_dis);
    for (int i = 0; i < m_max_dis; i++)
        m_object_disparity_range[i] = ComputeObjectDisparityRange((float)i); /* :111-115 */

    PrecomputeSky();
    PrecomputeObject();

    m_obj_cost_lut.resize((size_t)m_max_dis * m_max_dis); /* :122-129 */
    for (int fn = 0; fn < m_max_dis; fn++)
        for (int dis = 0; dis < m_max_dis; dis++)
            m_obj_cost_lut[(size_t)fn * m_max_dis + dis] = GetDataCostObject(fn, dis);

    const int rows_power2 = (int)powf(2, ceilf(log2f(m_rows + 1))); /* :131-133 */
    const int rows_power2_segmentation = (int)powf(2, ceilf(log2f(m_rows / 8 + 1)));

    /* kernel parameter block, Stixels.cu:212-245 */
    m_params.vhor = 0;
    m_params.rows = m_rows;
    m_params.cols = m_realcols;
    m_params.max_dis = m_max_dis;
    m_params.invalid_disparity = m_invalid_disparity;
    m_params.rows_log = m_rows_log;
    m_params.normalization_sky = m_normalization_sky;
    m_params.inv_sigma2_sky = m_inv_sigma2_sky;
    m_params.puniform_sky = m_puniform_sky;
    m_params.puniform = m_puniform;
    m_params.baseline = m_baseline;
    m_params.focal = m_focal;
    m_params.range_objects_z = m_range_objects_z;
    m_params.pord = m_pord;
    m_params.epsilon = m_epsilon;
    m_params.pgrav = m_pgrav;
    m_params.pblg = m_pblg;
    m_params.rows_power2 = rows_power2;
    m_params.rows_power2_segmentation = rows_power2_segmentation;
    m_params.max_sections = m_max_sections;
    m_params.max_dis_log = m_max_dis_log;
    m_params.width_margin = m_width_margin;
    m_params.segmentation_classes = m_segmentation_classes;
    m_params.segmentation_channels = m_segmentation_channels;
    m_params.prior_weight = m_prior_weight;
    m_params.disparity_weight = m_disparity_weight;
    m_params.segmentation_weight = m_segmentation_weight;
    m_params.instance_weight = m_instance_weight;
    m_params.column_step = m_column_step;
}

void Stixels::InitializeBatch(int max_batch) { /* Stixels.cu:43-248 */
    m_max_batch = std::max(1, max_batch);
    PrecomputeHost();
    const size_t inst_n = (size_t)m_instance_classes * m_realcols * m_max_sections;
    const int rows_power2_segmentation = m_params.rows_power2_segmentation;

    /* device side: LUT upload + all buffers (Stixels.cu:53-74, 136-210) on the caller's current
     * device unless SetDevice() chose one */
    int device = m_device;
    if (device < 0) IS_CHECK_RETURN(is_get_device(&device));
    m_ctx_device = device; /* (m_device keeps the REQUEST: -1 resolves again at the next Initialize) */
    const DeviceGuard guard(device);
    IS_CHECK_RETURN(is_ctx_create(&m_params, m_obj_cost_lut.data(),
                                  m_object_disparity_range.data(), m_max_batch, device, &m_ctx));
    IS_CHECK_RETURN(is_stream_create(&m_stream, 1));
    const size_t B = m_max_batch;
    /* [header rows: per-class counts][B x C x S sections]: one pitched copy fetches both */
    const size_t row_bytes = (size_t)m_max_sections * sizeof(Section);
    m_header_rows = (int)((B * m_instance_classes * sizeof(int32_t) + row_bytes - 1) / row_bytes);
    m_head_sections = m_max_sections < 64 ? m_max_sections : 64;
    IS_CHECK_RETURN(is_device_malloc((void**)&d_stixels_block,
                                     ((size_t)m_header_rows + B * m_realcols) * row_bytes));
    d_stixels = d_stixels_block + (size_t)m_header_rows * m_max_sections;
    d_instances_per_class = reinterpret_cast<int32_t*>(d_stixels_block);
    IS_CHECK_RETURN(is_device_malloc((void**)&d_instance_centerofmass,
                                     B * inst_n * 2 * sizeof(float)));
    IS_CHECK_RETURN(is_device_malloc((void**)&d_instance_indices,
                                     B * inst_n * 2 * sizeof(int32_t)));
    IS_CHECK_RETURN(is_device_malloc((void**)&d_instance_core_candidates, B * inst_n));
    IS_CHECK_RETURN(is_device_malloc(
        (void**)&d_segmentation,
        (size_t)rows_power2_segmentation * m_realcols * m_segmentation_channels * sizeof(int32_t)));
    IS_CHECK_RETURN(is_device_malloc((void**)&d_disparity_big,
                                     (size_t)m_rows * m_cols * sizeof(float)));
    IS_CHECK_RETURN(is_device_malloc((void**)&d_disparity,
                                     B * m_rows * m_realcols * sizeof(float)));
    IS_CHECK_RETURN(is_device_malloc((void**)&d_instance_labels, B * inst_n * sizeof(int32_t)));
    IS_CHECK_RETURN(is_device_malloc((void**)&d_instance_packed, B * (1 + 3 * inst_n) * sizeof(int32_t)));
    IS_CHECK_RETURN(is_host_malloc((void**)&h_stixels,
                                   (size_t)m_realcols * m_max_sections * sizeof(Section)));
    IS_CHECK_RETURN(is_host_malloc((void**)&h_stixels_head,
                                   ((size_t)m_header_rows + m_realcols) * m_head_sections * sizeof(Section)));
    IS_CHECK_RETURN(is_host_malloc((void**)&h_instance_head, B * 16 * sizeof(int32_t)));
    IS_CHECK_RETURN(is_host_malloc((void**)&h_instance_packed, (1 + 3 * inst_n) * sizeof(int32_t)));
    h_instance_packed[0] = 0;
    m_ground_valid = false; /* (the ground model depends on the configuration just applied) */
    m_is_initialized = true;
}

void Stixels::Finish() { /* Stixels.cu:250-283 */
    const DeviceGuard guard(m_ctx_device);
    IS_CHECK_RETURN(is_device_free(d_segmentation));
    IS_CHECK_RETURN(is_device_free(d_disparity_big));
    IS_CHECK_RETURN(is_device_free(d_disparity));
    IS_CHECK_RETURN(is_device_free(d_stixels_block));
    IS_CHECK_RETURN(is_device_free(d_instance_centerofmass));
    IS_CHECK_RETURN(is_device_free(d_instance_indices));
    IS_CHECK_RETURN(is_device_free(d_instance_core_candidates));
    IS_CHECK_RETURN(is_device_free(d_instance_labels));
    IS_CHECK_RETURN(is_device_free(d_instance_packed));
    IS_CHECK_RETURN(is_device_free(d_pack_counts));
    IS_CHECK_RETURN(is_device_free(d_pack_offsets));
    IS_CHECK_RETURN(is_device_free(d_pack_sections));
    IS_CHECK_RETURN(is_device_free(d_all_counts));
    IS_CHECK_RETURN(is_device_free(d_all_packed));
    IS_CHECK_RETURN(is_device_free(d_all_sections));
    d_pack_counts = d_pack_offsets = d_all_counts = nullptr;
    d_pack_sections = d_all_packed = d_all_sections = nullptr;
    m_all_columns_cap = m_all_packed_cap = 0;
    if (h_pack_offsets) IS_CHECK_RETURN(is_host_free(h_pack_offsets));
    if (h_pack_sections) IS_CHECK_RETURN(is_host_free(h_pack_sections));
    if (h_all_counts) IS_CHECK_RETURN(is_host_free(h_all_counts));
    h_pack_offsets = nullptr; h_pack_sections = nullptr; h_all_counts = nullptr;
    m_h_pack_cap = m_h_all_counts_cap = 0;
    IS_CHECK_RETURN(is_host_free(h_stixels));
    IS_CHECK_RETURN(is_host_free(h_stixels_head));
    h_stixels_head = nullptr; d_stixels_block = nullptr;
    IS_CHECK_RETURN(is_host_free(h_instance_head));
    IS_CHECK_RETURN(is_host_free(h_instance_packed));
    d_instance_labels = nullptr; d_instance_packed = nullptr;
    h_stixels = nullptr; h_instance_head = nullptr; h_instance_packed = nullptr;
    IS_CHECK_RETURN(is_ctx_destroy(m_ctx));
    m_ctx = nullptr;
    IS_CHECK_RETURN(is_stream_destroy(m_stream));
    m_stream = nullptr;
    d_segmentation = nullptr; d_disparity_big = nullptr; d_disparity = nullptr;
    d_stixels = nullptr; d_instance_centerofmass = nullptr; d_instance_indices = nullptr;
    d_instance_core_candidates = nullptr; d_instances_per_class = nullptr;
    m_log_lut.clear(); m_obj_cost_lut.clear(); m_object_disparity_range.clear();
    m_normalization_object.clear(); m_inv_sigma2_object.clear();
    m_is_initialized = false;
    m_ctx_device = -1;
}

/* ---------------------------------------------------------------- per-frame inputs */

void Stixels::SetSegmentation(const std::vector<int32_t>& segmentation) { /* :340-346 */
    const DeviceGuard guard(m_ctx_device);
    IS_CHECK_RETURN(is_memcpy_h2d(d_segmentation, segmentation.data(),
                                  sizeof(int32_t) * segmentation.size(), m_stream));
    IS_CHECK_RETURN(is_stream_synchronize(m_stream));
}

void Stixels::SetDisparityImage(const std::vector<pixel_t>& disp_im) { /* :348-355 */
    const DeviceGuard guard(m_ctx_device);
    /* the reference queues a cudaMemcpyAsync from the caller's pageable vector; the copy is
     * finished here before returning, so the vector may be a temporary */
    IS_CHECK_RETURN(is_memcpy_h2d(d_disparity_big, disp_im.data(),
                                  sizeof(pixel_t) * disp_im.size(), m_stream));
    IS_CHECK_RETURN(is_stream_synchronize(m_stream));
}

pixel_t* Stixels::GetInputDisparityImageOnDevice() { return d_disparity_big; } /* :357-359 */
int Stixels::GetRealCols() { return m_realcols; }
int Stixels::GetMaxSections() { return m_max_sections; }

/* ---------------------------------------------------------------- Compute */

void Stixels::FillHeader(StixelsData& d, float alpha_ground, int vhor_lib) const { /* :615-627 */
    d.sections.resize((size_t)m_realcols * m_max_sections);
    d.rows = m_rows;
    d.cols = m_cols;
    d.realcols = m_realcols;
    d.max_sections = m_max_sections;
    d.max_dis = m_max_dis;
    d.column_step = m_column_step;
    d.semantic_classes = m_segmentation_classes;
    d.alpha_ground = alpha_ground;
    d.vhor = vhor_lib;
}

float Stixels::Compute(const bool pairwise, StixelsData& stixels_data,
                       int32_t* d_segmentation_local) { /* Stixels.cu:449-637 */
    if (d_segmentation_local == nullptr) d_segmentation_local = d_segmentation;
    const DeviceGuard guard(m_ctx_device);

    /* JoinColumns does not need the ground model: it runs while the host computes it */
    IS_CHECK_RETURN(is_join_columns(m_ctx, d_disparity_big, m_cols, m_median_join ? 1 : 0,
                                    d_disparity, 1, m_stream)); /* :509-511 */
    GroundModel& g = m_ground;
    const float key[12] = {(float)m_vhor, m_camera_tilt, m_camera_height, m_alpha_ground, m_focal, m_baseline,
                           m_pout, m_sigma_camera_height, m_sigma_camera_tilt, m_sigma_disparity_ground,
                           m_max_disf, (float)m_rows};
    /* (memcmp: a NaN parameter compares equal to itself here, the model it gives is the same) */
    if (!m_ground_valid || std::memcmp(key, m_ground_key, sizeof(key)) != 0) {
        PrecomputeGround(m_vhor, m_camera_tilt, m_camera_height, m_alpha_ground, g); /* :463 */
        std::memcpy(m_ground_key, key, sizeof(key));
        m_ground_valid = true;
    }
    m_params.vhor = m_vhor;                                                       /* :532 */
    /* the DP, the instance candidates and their clustering (ClusterInstances, :613) are queued
     * back to back on the device; nothing returns to the host in between */
    const is_instance_buffers ib = InstanceBuffers(0);
    IS_CHECK_RETURN(is_compute(m_ctx, d_disparity, d_segmentation_local, g.function.data(),
                               g.normalization.data(), g.inv_sigma2.data(), &m_vhor,
                               pairwise ? 1 : 0, 1, d_stixels, &ib, nullptr, nullptr,
                               m_stream)); /* :535-590 */
    /* results into pinned memory, ONE copy and ONE synchronisation (:600, :629-633): the header
     * row(s) with the per-class counts and the first m_head_sections sections of every column (a
     * column rarely has more: 10-40 on road scenes) */
    const int K = m_head_sections;
    const size_t row_bytes = (size_t)m_max_sections * sizeof(Section);
    IS_CHECK_RETURN(is_memcpy2d_d2h(h_stixels_head, (size_t)K * sizeof(Section), d_stixels_block, row_bytes,
                                    (size_t)K * sizeof(Section), (size_t)m_header_rows + m_realcols,
                                    m_stream));
    IS_CHECK_RETURN(is_stream_synchronize(m_stream));
    const int32_t* head = reinterpret_cast<const int32_t*>(h_stixels_head);
    for (int k = 0; k < m_instance_classes; k++) m_instances_per_class[k] = head[k];
    m_labels_on_host = false;

    FillHeader(stixels_data, m_alpha_ground, m_vhor);
    /* sections of every column up to and including its terminator; what lies behind a
     * terminator is unspecified (in the reference: whatever the device buffer held) */
    Section* out = stixels_data.sections.data();
    const Section* cols = h_stixels_head + (size_t)m_header_rows * K;
    bool complete = true;
    for (int c = 0; c < m_realcols && complete; c++) {
        const Section* src = cols + (size_t)c * K;
        int n = 0;
        while (n < K && src[n].type != -1) n++;
        if (n == K && K < m_max_sections) { complete = false; break; } /* no terminator among the first K */
        if (n == K) n = K - 1; /* (K == max_sections: the last slot ends the column, as below) */
        std::memcpy(out + (size_t)c * m_max_sections, src, (size_t)(n + 1) * sizeof(Section));
    }
    if (!complete) { /* a column with more than K sections: fetch everything */
        const size_t n_sec = (size_t)m_realcols * m_max_sections;
        IS_CHECK_RETURN(is_memcpy_d2h(h_stixels, d_stixels, n_sec * sizeof(Section), m_stream));
        IS_CHECK_RETURN(is_stream_synchronize(m_stream));
        for (int c = 0; c < m_realcols; c++) {
            const Section* src = h_stixels + (size_t)c * m_max_sections;
            int n = 0;
            while (n < m_max_sections - 1 && src[n].type != -1) n++;
            std::memcpy(out + (size_t)c * m_max_sections, src, (size_t)(n + 1) * sizeof(Section));
        }
    }
    return -1; /* the reference's timers are commented out, Stixels.cu:636 */
}

is_instance_buffers Stixels::InstanceBuffers(int image) const {
    const size_t inst_n = (size_t)m_instance_classes * m_realcols * m_max_sections;
    const size_t i = (size_t)image;
    is_instance_buffers ib = {}; /* (zero-initialised: fields added by later versions stay NULL) */
    ib.d_centerofmass = d_instance_centerofmass + i * inst_n * 2;
    ib.d_indices = d_instance_indices + i * inst_n * 2;
    ib.d_core_candidates = d_instance_core_candidates + i * inst_n;
    ib.d_instances_per_class = d_instances_per_class + i * m_instance_classes;
    ib.d_labels = d_instance_labels + i * inst_n;
    ib.d_packed = d_instance_packed + i * (1 + 3 * inst_n);
    return ib;
}

void Stixels::ComputeBatch(bool pairwise, int n_images, const pixel_t* d_big,
                           const int32_t* d_seg, const RoadParameters* road,
                           std::vector<StixelsData>& out, void* stream,
                           std::vector<InstanceMapping>* instance_stixels) {
    if (n_images < 1 || n_images > m_max_batch)
        throw std::invalid_argument("n_images outside [1, max_batch] of InitializeBatch().");
    const DeviceGuard guard(m_ctx_device);
    if (stream == nullptr) stream = m_stream;
    std::vector<float> gf((size_t)n_images * m_rows), ng(gf.size()), ig(gf.size());
    std::vector<int> vh(n_images);
    for (int i = 0; i < n_images; i++) {
        GroundModel g;
        vh[i] = m_rows - road[i].vhor - 1;
        PrecomputeGround(vh[i], road[i].camera_tilt, road[i].camera_height, road[i].alpha_ground,
                         g);
        std::copy(g.function.begin(), g.function.end(), gf.begin() + (size_t)i * m_rows);
        std::copy(g.normalization.begin(), g.normalization.end(), ng.begin() + (size_t)i * m_rows);
        std::copy(g.inv_sigma2.begin(), g.inv_sigma2.end(), ig.begin() + (size_t)i * m_rows);
    }
    IS_CHECK_RETURN(is_join_columns(m_ctx, d_big, m_cols, m_median_join ? 1 : 0, d_disparity,
                                    n_images, stream));
    /* instance candidates + clustering of every frame: two more launches for the whole batch */
    std::vector<is_instance_buffers> ibs;
    if (instance_stixels)
        for (int i = 0; i < n_images; i++) ibs.push_back(InstanceBuffers(i));
    IS_CHECK_RETURN(is_compute(m_ctx, d_disparity, d_seg, gf.data(), ng.data(), ig.data(),
                               vh.data(), pairwise ? 1 : 0, n_images, d_stixels,
                               instance_stixels ? ibs.data() : nullptr, nullptr, nullptr, stream));
    /* Results to the host COMPACTED and through pinned memory: a column uses 10-60 of its 200 slots, and the
     * reference's fixed-stride copy (Stixels.cu:629-633: one frame) would move 1.6 MB per frame into pageable
     * vectors.  is_pack_sections leaves per-column offsets + the used sections; two pinned copies (the offsets, then
     * exactly the used sections) and a scatter on the host restore the fixed-stride layout incl. each column's
     * terminator -- what lies behind a terminator is unspecified, as in Compute(). */
    const size_t ncols = (size_t)n_images * m_realcols;
    EnsurePackBuffers();
    IS_CHECK_RETURN(is_pack_sections((const is_section*)d_stixels, (int)ncols, m_max_sections, d_pack_counts,
                                     d_pack_offsets, (is_section*)d_pack_sections, stream));
    IS_CHECK_RETURN(is_memcpy_d2h(h_pack_offsets, d_pack_offsets, (ncols + 1) * sizeof(int32_t), stream));
    if (instance_stixels) /* the per-class counts of all frames: one small copy */
        IS_CHECK_RETURN(is_memcpy_d2h(h_instance_head, d_instances_per_class,
                                      (size_t)n_images * m_instance_classes * sizeof(int32_t), stream));
    IS_CHECK_RETURN(is_stream_synchronize(stream));
    const size_t total = (size_t)h_pack_offsets[ncols];
    if (total > m_h_pack_cap) { /* (first call: 64 per column; grown to what a batch needs) */
        if (h_pack_sections) IS_CHECK_RETURN(is_host_free(h_pack_sections));
        h_pack_sections = nullptr;
        m_h_pack_cap = total + total / 4 + 1024;
        IS_CHECK_RETURN(is_host_malloc((void**)&h_pack_sections, m_h_pack_cap * sizeof(Section)));
    }
    if (total > 0)
        IS_CHECK_RETURN(is_memcpy_d2h(h_pack_sections, d_pack_sections, total * sizeof(Section), stream));
    out.resize(n_images);
    for (int i = 0; i < n_images; i++) FillHeader(out[i], road[i].alpha_ground, vh[i]); /* (beside the copy) */
    IS_CHECK_RETURN(is_stream_synchronize(stream));
    Section term;
    std::memset(&term, 0, sizeof(term));
    term.type = -1; /* StixelsKernels.cu:952-954 */
    for (int i = 0; i < n_images; i++) {
        Section* dst = out[i].sections.data();
        for (int c = 0; c < m_realcols; c++) {
            const size_t col = (size_t)i * m_realcols + c;
            const size_t o = (size_t)h_pack_offsets[col], n = (size_t)h_pack_offsets[col + 1] - o;
            if (n > 0) std::memcpy(dst + (size_t)c * m_max_sections, h_pack_sections + o, n * sizeof(Section));
            dst[(size_t)c * m_max_sections + n] = term;
        }
    }
    if (!instance_stixels) return;
    /* (column, section, label) triples of every frame, sized by the counts just read */
    const size_t inst_n = (size_t)m_instance_classes * m_realcols * m_max_sections;
    std::vector<int> totals(n_images, 0);
    std::vector<std::vector<int32_t>> triples(n_images);
    for (int i = 0; i < n_images; i++) {
        for (int k = 0; k < m_instance_classes; k++) totals[i] += h_instance_head[i * m_instance_classes + k];
        triples[i].resize(3 * (size_t)totals[i] + 1);
        if (totals[i] > 0)
            IS_CHECK_RETURN(is_memcpy_d2h(triples[i].data(), d_instance_packed + (size_t)i * (1 + 3 * inst_n),
                                          (1 + 3 * (size_t)totals[i]) * sizeof(int32_t), stream));
    }
    IS_CHECK_RETURN(is_stream_synchronize(stream));
    instance_stixels->assign(n_images, InstanceMapping());
    for (int i = 0; i < n_images; i++) {
        const int32_t* t = triples[i].data() + 1;
        for (int j = 0; j < totals[i]; j++)
            (*instance_stixels)[i][std::make_pair(t[3 * j], t[3 * j + 1])] = t[3 * j + 2];
    }
    /* a following GetInstanceStixels() returns the mapping of frame 0 (slice 0 of the arrays) */
    for (int k = 0; k < m_instance_classes; k++) m_instances_per_class[k] = h_instance_head[k];
    m_labels_on_host = false;
}

/* the packed payload of a batch (ComputeBatch, ComputeBatchGather): allocated on first use, released by Finish */
void Stixels::EnsurePackBuffers() {
    if (d_pack_counts != nullptr) return;
    const size_t cols = (size_t)m_max_batch * m_realcols;
    IS_CHECK_RETURN(is_device_malloc((void**)&d_pack_counts, cols * sizeof(int32_t)));
    IS_CHECK_RETURN(is_device_malloc((void**)&d_pack_offsets, (cols + 1) * sizeof(int32_t)));
    IS_CHECK_RETURN(is_device_malloc((void**)&d_pack_sections, cols * (m_max_sections - 1) * sizeof(Section)));
    IS_CHECK_RETURN(is_host_malloc((void**)&h_pack_offsets, (cols + 1) * sizeof(int32_t)));
    if (h_pack_sections != nullptr) return; /* (ComputeBatchGather on dst may have made it already) */
    m_h_pack_cap = cols * 64;
    IS_CHECK_RETURN(is_host_malloc((void**)&h_pack_sections, m_h_pack_cap * sizeof(Section)));
}

/* The shard of this rank, then the compacted gather of every rank's Sections on `dst` (the
 * C ABI underneath: is_pack_sections -> is_gather_sections -> is_unpack_sections). */
void Stixels::ComputeBatchGather(bool pairwise, int n_images, const pixel_t* d_big, const int32_t* d_seg,
                                 const RoadParameters* road, void* comm, int dst, const int* images_per_rank,
                                 const RoadParameters* road_all, std::vector<StixelsData>& out, void* stream) {
    if (n_images < 1 || n_images > m_max_batch)
        throw std::invalid_argument("n_images outside [1, max_batch] of InitializeBatch().");
    if (comm == nullptr || images_per_rank == nullptr)
        throw std::invalid_argument("ComputeBatchGather needs a communicator and the shard sizes.");
    const DeviceGuard guard(m_ctx_device);
    if (stream == nullptr) stream = m_stream;
    int rank = 0, nranks = 0;
    IS_CHECK_RETURN(is_comm_rank(comm, &rank, &nranks));
    if (images_per_rank[rank] != n_images)
        throw std::invalid_argument("images_per_rank[rank] differs from n_images.");
    if (rank == dst && road_all == nullptr)
        throw std::invalid_argument("the destination rank needs road_all (the headers of every frame).");
    const size_t per = (size_t)m_realcols * m_max_sections;
    const int my_cols = n_images * m_realcols;

    /* ---- this rank's shard: ground model on the host, JoinColumns + DP + back-trace on the device */
    std::vector<float> gf((size_t)n_images * m_rows), ng(gf.size()), ig(gf.size());
    std::vector<int> vh(n_images);
    for (int i = 0; i < n_images; i++) {
        GroundModel g;
        vh[i] = m_rows - road[i].vhor - 1;
        PrecomputeGround(vh[i], road[i].camera_tilt, road[i].camera_height, road[i].alpha_ground, g);
        std::copy(g.function.begin(), g.function.end(), gf.begin() + (size_t)i * m_rows);
        std::copy(g.normalization.begin(), g.normalization.end(), ng.begin() + (size_t)i * m_rows);
        std::copy(g.inv_sigma2.begin(), g.inv_sigma2.end(), ig.begin() + (size_t)i * m_rows);
    }
    IS_CHECK_RETURN(is_join_columns(m_ctx, d_big, m_cols, m_median_join ? 1 : 0, d_disparity, n_images, stream));
    IS_CHECK_RETURN(is_compute(m_ctx, d_disparity, d_seg, gf.data(), ng.data(), ig.data(), vh.data(),
                               pairwise ? 1 : 0, n_images, d_stixels, nullptr, nullptr, nullptr, stream));

    /* ---- pack: per-column counts + the used sections (10-40 of the 200 slots of a column) */
    EnsurePackBuffers();
    IS_CHECK_RETURN(is_pack_sections((const is_section*)d_stixels, my_cols, m_max_sections, d_pack_counts,
                                     d_pack_offsets, (is_section*)d_pack_sections, stream));

    /* ---- gather on dst.  Landing buffers: counts for every column, sections for 64 per column at first
     * (synthetic and real scenes use 10-60); when a batch needs more, EVERY rank sees IS_ENOMEM from the
     * gather (dst's go-ahead), dst grows to the worst case and all ranks repeat the call */
    std::vector<int32_t> columns(nranks);
    size_t all_cols = 0;
    int all_images = 0;
    for (int r = 0; r < nranks; r++) {
        columns[r] = images_per_rank[r] * m_realcols;
        all_cols += (size_t)columns[r];
        all_images += images_per_rank[r];
    }
    std::vector<int64_t> totals(nranks, 0);
    if (rank == dst && (m_all_columns_cap < all_cols || d_all_counts == nullptr)) {
        IS_CHECK_RETURN(is_stream_synchronize(stream));
        IS_CHECK_RETURN(is_device_free(d_all_counts));
        IS_CHECK_RETURN(is_device_malloc((void**)&d_all_counts, (all_cols + 1) * sizeof(int32_t) * 2));
        m_all_columns_cap = all_cols;
    }
    for (int attempt = 0;; attempt++) {
        if (rank == dst) {
            const size_t want = attempt == 0 ? all_cols * 64 : all_cols * (size_t)(m_max_sections - 1);
            if (m_all_packed_cap < want) {
                IS_CHECK_RETURN(is_stream_synchronize(stream));
                IS_CHECK_RETURN(is_device_free(d_all_packed));
                IS_CHECK_RETURN(is_device_malloc((void**)&d_all_packed, want * sizeof(Section)));
                m_all_packed_cap = want;
            }
        }
        const int rc = is_gather_sections(comm, dst, columns.data(), d_pack_counts, d_pack_offsets,
                                          (const is_section*)d_pack_sections, d_all_counts,
                                          (is_section*)d_all_packed, m_all_packed_cap, totals.data(), stream);
        if (rc == IS_ENOMEM && attempt == 0) continue; /* (every rank takes this branch together) */
        IS_CHECK_RETURN(rc);
        break;
    }
    out.clear();
    if (rank != dst) {
        IS_CHECK_RETURN(is_stream_synchronize(stream)); /* the payload has left before the buffers are reused */
        return;
    }
    /* ---- dst: the packed payload of all ranks to the host through pinned memory (per-column counts + exactly the used
     * sections), then back to fixed-stride Section arrays with each column's terminator on the host -- like
     * ComputeBatch; is_unpack_sections is the device-side form of the same scatter for callers that keep the result
     * on the GPU */
    size_t total = 0;
    for (int r = 0; r < nranks; r++) total += (size_t)totals[r];
    if (m_h_all_counts_cap < all_cols) {
        if (h_all_counts) IS_CHECK_RETURN(is_host_free(h_all_counts));
        h_all_counts = nullptr;
        IS_CHECK_RETURN(is_host_malloc((void**)&h_all_counts, all_cols * sizeof(int32_t)));
        m_h_all_counts_cap = all_cols;
    }
    if (total > m_h_pack_cap) {
        if (h_pack_sections) IS_CHECK_RETURN(is_host_free(h_pack_sections));
        h_pack_sections = nullptr;
        m_h_pack_cap = total + total / 4 + 1024;
        IS_CHECK_RETURN(is_host_malloc((void**)&h_pack_sections, m_h_pack_cap * sizeof(Section)));
    }
    IS_CHECK_RETURN(is_memcpy_d2h(h_all_counts, d_all_counts, all_cols * sizeof(int32_t), stream));
    if (total > 0)
        IS_CHECK_RETURN(is_memcpy_d2h(h_pack_sections, d_all_packed, total * sizeof(Section), stream));
    out.resize(all_images);
    for (int i = 0; i < all_images; i++) FillHeader(out[i], road_all[i].alpha_ground, m_rows - road_all[i].vhor - 1);
    IS_CHECK_RETURN(is_stream_synchronize(stream));
    Section term;
    std::memset(&term, 0, sizeof(term));
    term.type = -1; /* StixelsKernels.cu:952-954 */
    size_t o = 0;
    for (int i = 0; i < all_images; i++) {
        Section* dst_sec = out[i].sections.data();
        for (int c = 0; c < m_realcols; c++) {
            int32_t n = h_all_counts[(size_t)i * m_realcols + c];
            n = n < 0 ? 0 : (n > m_max_sections - 1 ? m_max_sections - 1 : n);
            if (o + (size_t)n > total) throw std::runtime_error("ComputeBatchGather: the gathered counts exceed the gathered sections.");
            if (n > 0) std::memcpy(dst_sec + (size_t)c * m_max_sections, h_pack_sections + o, (size_t)n * sizeof(Section));
            dst_sec[(size_t)c * m_max_sections + n] = term;
            o += (size_t)n;
        }
    }
    (void)per;
}

/* ---------------------------------------------------------------- instances */

/* Size-filtered DBSCAN over the predicted instance centres of each instance class, on the
 * device (k_cluster_instances, csrc/is_k_cluster.hip).  The reference calls a cuML fork whose
 * source is not in its tree (Stixels.cu:639-681); the semantics are those of
 * its Python twin (/root/reference/tools/visualization/clustering_visualization.py:894-960).
 * Compute() already runs it; calling it again re-clusters the candidates of the last frame with
 * the eps / min_pts of Initialize() (m_params, like the reference). */
float Stixels::ClusterInstances() {
    const DeviceGuard guard(m_ctx_device);
    const is_instance_buffers ib = InstanceBuffers();
    IS_CHECK_RETURN(is_cluster_instances(m_ctx, &ib, m_stream));
    m_labels_on_host = false;
    return -1;
}

std::map<std::pair<int, int>, int> Stixels::GetInstanceStixels() { /* Stixels.cu:744-776 */
    /* the reference copies the complete label and index arrays (4.8 MB, "~0.8 milliseconds",
     * :745); here the device has packed (column, section, label) triples of the candidates */
    if (!m_labels_on_host) {
        const DeviceGuard guard(m_ctx_device);
        int total = 0;
        for (int k = 0; k < m_instance_classes; k++) total += m_instances_per_class[k];
        if (total > 0) {
            IS_CHECK_RETURN(is_memcpy_d2h(h_instance_packed, d_instance_packed,
                                          (1 + 3 * (size_t)total) * sizeof(int32_t), m_stream));
            IS_CHECK_RETURN(is_stream_synchronize(m_stream));
        } else {
            h_instance_packed[0] = 0;
        }
        m_labels_on_host = true;
    }
    std::map<std::pair<int, int>, int> mapping;
    const int total = h_instance_packed[0];
    const int32_t* t = h_instance_packed + 1;
    for (int i = 0; i < total; i++)
        mapping[std::make_pair(t[3 * i], t[3 * i + 1])] = t[3 * i + 2];
    return mapping;
}

/* ---------------------------------------------------------------- outputs */

std::vector<float> Stixels::Get3DVertices(const StixelsData& stixels_data) { /* :683-742 */
    if (m_camera_center_x == -1 || m_camera_center_y == -1)
        throw std::invalid_argument("Camera parameters are not set.");
    std::vector<float> vertices;
    for (size_t i = 0; i < (size_t)m_realcols; i++) {
        for (size_t j = 0; j < (size_t)m_max_sections; j++) {
            const Section& section = stixels_data.sections[i * m_max_sections + j];
            if (section.type == -1) break;
            const float x_l = i * m_column_step;
            const float x_r = x_l + m_column_step;
            const float y_t = m_rows - section.vT - 1;
            const float y_b = m_rows - section.vB;
            float top_depth = 0.0, bottom_depth = 0.0; /* sky stays at depth 0 */
            if (section.type == OBJECT) {
                top_depth = m_baseline * m_focal / section.disparity;
                bottom_depth = top_depth;
            } else if (section.type == GROUND) {
                const float top_disparity =
                    stixels_data.alpha_ground * (stixels_data.vhor - section.vT);
                const float bottom_disparity =
                    stixels_data.alpha_ground * (stixels_data.vhor - section.vB);
                top_depth = m_baseline * m_focal / top_disparity;
                bottom_depth = m_baseline * m_focal / bottom_disparity;
            }
            const float corner[4][3] = {
                {x_l, y_t, top_depth}, {x_r, y_t, top_depth},
                {x_r, y_b, bottom_depth}, {x_l, y_b, bottom_depth}}; /* clockwise from top left */
            for (const auto& c : corner) {
                vertices.push_back(-c[2] / m_focal * (m_camera_center_x - c[0]));
                vertices.push_back(-c[2] / m_focal * (m_camera_center_y - c[1]));
                vertices.push_back(c[2]);
            }
        }
    }
    return vertices;
}

void Stixels::SaveStixels(Section* stixels,
                          std::map<std::pair<int, int>, int> instance_stixels_mapping,
                          const float alpha_ground, const int vhor, const int real_cols,
                          const int max_segments, const char* fname) { /* Stixels.cu:889-926 */
    std::ofstream fp;
    fp.open(fname, std::ofstream::out | std::ofstream::trunc);
    if (!fp.is_open()) {
        std::cerr << "Counldn't write file: " << fname << std::endl;
        return;
    }
    for (size_t i = 0; i < (size_t)real_cols; i++) {
        for (size_t j = 0; j < (size_t)max_segments; j++) {
            const Section& section = stixels[i * max_segments + j];
            if (section.type == -1) break;
            fp << section.type << "," << section.vB << "," << section.vT << ","
               << section.disparity << "," << section.semantic_class << "," << section.cost << ","
               << section.instance_meanx << "," << section.instance_meany;
            const auto it = instance_stixels_mapping.find(std::make_pair((int)i, (int)j));
            if (it != instance_stixels_mapping.end()) fp << "," << (*it).second;
            fp << ";";
        }
        fp << std::endl;
    }
    fp << "groundplane" << alpha_ground << "," << vhor << "\n";
    fp.close();
}
