/*
 * types.h -- boundary types of the stixel library, source-compatible with
 * /root/reference/InstanceStixels/include/InstanceStixels/types.h:22-205.
 *
 * StixelParameters and Section ARE the C structs of the HIP core's ABI (same field names,
 * order and layout as types.h:145-194), so a StixelsData filled by this library can be
 * consumed by the reference's callers unchanged.
 */
#ifndef INSTANCESTIXELS_AMD_TYPES_H_
#define INSTANCESTIXELS_AMD_TYPES_H_

#include <vector>

#include "instance_stixels_core.h"

constexpr int GROUND = IS_GROUND;
constexpr int OBJECT = IS_OBJECT;
constexpr int SKY = IS_SKY;

typedef is_stixel_params StixelParameters; /* types.h:145-184 */
typedef is_section Section;                /* types.h:186-194 */

/* User-facing configuration; mandatory fields default to -1 and are checked by
 * Stixels::SetConfig (types.h:30-141, Stixels.cu:292-313). */
struct StixelConfig {
    /* mandatory: image, clustering, CNN, weights, stixel width, camera */
    float rows = -1, cols = -1;
    int max_dis = -1;
    float invalid_disparity = -1.0f; /* < 0: no explicit invalid value; else usually 0 */
    float eps = -1;
    int min_pts = -1, size_filter = -1;
    int n_semantic_classes = -1, n_offset_channels = -1;
    float prior_weight = -1, segmentation_weight = -1, instance_weight = -1, disparity_weight = -1;
    bool pairwise = false; /* convenience storage only; passed to Compute() */
    int column_step = -1;
    float focal = -1, baseline = -1, camera_center_x = -1, camera_center_y = -1;

    /* optional: disparity model */
    float sigma_disparity_object = 1.0f, sigma_disparity_ground = 2.0f, sigma_sky = 0.1f;
    /* optional: probabilities */
    float pout = 0.15f, pout_sky = 0.4f, pord = 0.2f, pgrav = 0.1f, pblg = 0.04f;
    float pground_given_nexist = 0.28, pobject_given_nexist = 0.44, psky_given_nexist = 0.28;
    float pnexist_dis = 0.25f;
    float pground = 1.0f / 3.0f, pobject = 1.0f / 3.0f, psky = 1.0f / 3.0f;
    /* optional: geometry */
    int width_margin = 0;
    float sigma_camera_tilt = 0.05f, sigma_camera_height = 0.05f;
    bool median_join = false;
    float epsilon = 3.0f;
    float range_objects_z = 10.20f;
    float road_vdisparity_threshold = 0.2f;
};

struct StixelsData { /* types.h:196-205 */
    std::vector<Section> sections; /* [realcols][max_sections], terminator type = -1 */
    int rows, cols;
    int realcols, max_sections, max_dis;
    int column_step;
    int semantic_classes;
    float alpha_ground;
    int vhor;
};

#endif
