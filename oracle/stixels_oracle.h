/*
 * stixels_oracle.h -- CPU restatement of the reference's column-DP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (instance_stixels_amd/, include/) may
 * include, link or call this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do.
 *
 * PARITY PIN STATUS: "parity unpinned" beyond the two known-answer patterns that the
 * reference's own (disabled) unit tests hold -- the exclusive-scan identity and the
 * column-join layout (InstanceStixels/tests/generate_testdata.py:51-62), see
 * tests/test_oracle_kat.py.  The reference is CUDA (no nvcc, no NVIDIA GPU in this image) and
 * building it needs stand-ins for CUDA/cuML headers, so it is treated as unbuildable here;
 * its end-to-end regression (tests/run_test.sh) needs Cityscapes + weights.  Every function
 * below therefore cites the reference file:line it restates so it can be audited by reading.
 */
#ifndef STIXELS_ORACLE_H_
#define STIXELS_ORACLE_H_

#include <stdint.h>
#include "instance_stixels_core.h" /* is_stixel_params, is_section (layout-only) */

#ifdef __cplusplus
extern "C" {
#endif

/* Mirror of `struct StixelConfig` (types.h:30-141), C types only. */
typedef struct orc_config {
    int rows, cols, max_dis;
    float invalid_disparity;
    float eps;
    int min_pts, size_filter;
    int n_semantic_classes, n_offset_channels;
    float prior_weight, segmentation_weight, instance_weight, disparity_weight;
    int column_step;
    float focal, baseline, camera_center_x, camera_center_y;
    float sigma_disparity_object, sigma_disparity_ground, sigma_sky;
    float pout, pout_sky, pord, pgrav, pblg;
    float pground_given_nexist, pobject_given_nexist, psky_given_nexist;
    float pnexist_dis, pground, pobject, psky;
    int width_margin;
    float sigma_camera_tilt, sigma_camera_height;
    int median_join;
    float epsilon, range_objects_z;
} orc_config;

void orc_default_config(orc_config* c);

/* Stixels::SetConfig + Initialize host precompute (Stixels.cu:43-248, 292-447, 819-887).
 * obj_cost_lut: [max_dis*max_dis], obj_disparity_range: [max_dis]. */
int orc_host_initialize(const orc_config* c, is_stixel_params* params, float* obj_cost_lut,
                        float* obj_disparity_range);

/* Stixels::SetRoadParameters + PrecomputeGround (Stixels.cu:375-381, 790-817, 867-877).
 * Outputs three arrays of `rows` floats and the library-convention horizon. */
int orc_host_ground(const orc_config* c, int vhor_image, float camera_tilt, float camera_height,
                    float alpha_ground, float* ground_function, float* normalization_ground,
                    float* inv_sigma2_ground, int* vhor_lib);

/* JoinColumns (StixelsKernels.cu:980-1095). disp_big [rows][cols] -> out [realcols][rows]. */
void orc_join_columns(const float* disp_big, float* out, int step_size, int median,
                      int width_margin, int rows, int cols, int real_cols,
                      float invalid_disparity);

/* ComputePrefixSum<T> (StixelsKernels.h:73-103), n a power of two, in place, exclusive. */
void orc_blelloch_f32(float* arr, int n);
void orc_blelloch_i32(int32_t* arr, int n);
void orc_blelloch_i64(int64_t* arr, int n);

/* ComputeObjectLUT for one column (StixelsKernels.cu:236-296, 959-978).
 * lut: [max_dis][rows_power2+1]. */
void orc_object_lut_column(const float* disp_col, const float* obj_cost_lut, float* lut,
                           const is_stixel_params* p, int n_power2);

/* ComputeObjectLUT + StixelsKernel<PAIRWISE> (Stixels.cu:535-590, StixelsKernels.cu:298-957)
 * for columns [col_begin, col_end) of one image, `nthreads` OpenMP threads.
 *   disp_joined [realcols][rows]; seg [realcols][channels][P2S] (NOT modified);
 *   sections [realcols][max_sections];
 *   cost_table (optional) [realcols][rows][3]; index_table (optional) [realcols][rows][3]
 *   (entries never written by the reference are reported as -1).
 * Instance candidates in canonical (column, section) order, reference layout
 * (class_offset = class*realcols*max_sections); any of them may be NULL. */
int orc_compute(const is_stixel_params* p, const float* obj_cost_lut,
                const float* obj_disparity_range, const float* disp_joined, const int32_t* seg,
                const float* ground_function, const float* normalization_ground,
                const float* inv_sigma2_ground, int pairwise, int col_begin, int col_end,
                int nthreads, is_section* sections, float* cost_table, int32_t* index_table,
                float* inst_centerofmass, int32_t* inst_indices, uint8_t* inst_core,
                int32_t* inst_per_class);

/* FlipAndPad, tools/CNN_training/models/wrappers.py:35-61: in [CH][Hs][Ws] float ->
 * out [Ws][CH][P2S] int32 (permute, flip rows, zero pad, (int)(8*x)). */
void orc_flip_and_pad(const float* in, int32_t* out, int CH, int Hs, int Ws, int P2S);

/* ComputeHistogram / ComputeMaximum / ComputeBinaryImage (RoadEstimationKernels.cu:25-60):
 * vdisp [rows][max_dis] int, binary [rows][max_dis] uint8; returns the maximum. */
int orc_road_vdisparity(const float* disparity, int rows, int cols, int max_dis, float threshold,
                        int* vdisp, uint8_t* binary);

float orc_logf(float x); /* = is_logf, exported for tests */

#ifdef __cplusplus
}
#endif
#endif
