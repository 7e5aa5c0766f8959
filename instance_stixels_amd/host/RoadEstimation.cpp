/*
 * RoadEstimation.cpp -- host class of the road estimation step, the
 * producer of the four scalars Stixels::SetRoadParameters consumes.  Mirrors
 * /root/reference/InstanceStixels/src/RoadEstimation.cu:32-193; citations `RE.cu:N` refer to it.
 * Device work goes through the C ABI (is_road_vdisparity); the line fit runs on the host like
 * the reference's cv::HoughLines call, with an own implementation of the standard transform.
 */
#include "InstanceStixels/RoadEstimation.h"
#include "DeviceGuard.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

static const float kPi = 3.1415926535897932384626433832795f; /* CV_PI as float */

RoadEstimation::RoadEstimation() {}
RoadEstimation::~RoadEstimation() {}

void RoadEstimation::Initialize(const float camera_center_y, const float baseline,
                                const float focal, const int rows, const int cols,
                                const int max_dis, const float road_vdisparity_threshold) {
    m_cy = camera_center_y; /* RE.cu:37-40 */
    m_b = baseline;
    /* re-initialisation (a new frame shape or camera, stixels_wrapper.cu:124-152) releases the stream and the
     * buffers of the previous one first -- on THEIR device, which Finish() still knows */
    if (m_is_initialized || m_stream) Finish();
    m_focal = focal;
    m_HoughAccumThr = 25; /* RE.cu:45-57 */
    m_binThr = road_vdisparity_threshold;
    m_maxPitch = 50 * kPi / 180.0f;
    m_minPitch = -50 * kPi / 180.0f;
    m_maxCameraHeight = 1.90f;
    m_minCameraHeight = 1.30f;
    m_max_dis = max_dis;
    m_rows = rows;
    m_cols = cols;
    m_rho = m_theta = 0;
    m_horizonPoint = 0;
    m_pitch = m_cameraHeight = 0;
    m_vDisp.assign((size_t)m_max_dis * m_rows, 0);
    /* the device of the buffers: SetDevice(), else the caller's current one (like Stixels::Initialize) */
    int device = m_device;
    if (device < 0) IS_CHECK_RETURN(is_get_device(&device));
    m_ctx_device = device;
    const DeviceGuard guard(device);
    IS_CHECK_RETURN(is_stream_create(&m_stream, 1));
    IS_CHECK_RETURN(is_device_malloc((void**)&d_disparity, (size_t)m_cols * m_rows * sizeof(float)));
    IS_CHECK_RETURN(is_device_malloc((void**)&d_vDisp, (size_t)m_max_dis * m_rows * sizeof(int)));
    IS_CHECK_RETURN(is_device_malloc((void**)&d_maximum, sizeof(int)));
    IS_CHECK_RETURN(is_device_malloc((void**)&d_vDispBinary, (size_t)m_max_dis * m_rows));
    m_is_initialized = true;
}

void RoadEstimation::Finish() { /* RE.cu:84-92 */
    const DeviceGuard guard(m_ctx_device);
    if (m_stream) {
        IS_CHECK_RETURN(is_stream_synchronize(m_stream));
        IS_CHECK_RETURN(is_stream_destroy(m_stream));
        m_stream = nullptr;
    }
    IS_CHECK_RETURN(is_device_free(d_vDisp));
    IS_CHECK_RETURN(is_device_free(d_disparity));
    IS_CHECK_RETURN(is_device_free(d_maximum));
    IS_CHECK_RETURN(is_device_free(d_vDispBinary));
    d_vDisp = nullptr; d_disparity = nullptr; d_maximum = nullptr; d_vDispBinary = nullptr;
    m_is_initialized = false;
}

bool RoadEstimation::Compute(const std::vector<pixel_t>& im) { /* RE.cu:94-102 */
    {
        const DeviceGuard guard(m_ctx_device);
        IS_CHECK_RETURN(is_memcpy_h2d(d_disparity, im.data(), im.size() * sizeof(pixel_t), m_stream));
        IS_CHECK_RETURN(is_stream_synchronize(m_stream)); /* the caller's vector may be a temporary */
    }
    return Compute(d_disparity);
}

bool RoadEstimation::Compute(pixel_t* d_im) { /* RE.cu:104-138 */
    const DeviceGuard guard(m_ctx_device); /* (d_im lives on the object's device: Stixels::SetDevice(d) + SetDevice(d)) */
    IS_CHECK_RETURN(is_road_vdisparity(d_im, m_rows, m_cols, m_max_dis, m_binThr, d_vDisp, d_maximum,
                                       d_vDispBinary, m_stream));
    float rho, theta, horizonPoint, pitch, cameraHeight, slope;
    bool ok = false;
    if (ComputeHough(rho, theta, horizonPoint, pitch, cameraHeight, slope)) {
        m_rho = rho;
        m_theta = theta;
        m_horizonPoint = (int)ceil(horizonPoint);
        m_pitch = pitch;
        m_cameraHeight = cameraHeight;
        m_slope = slope;
        ok = true;
    }
    return ok;
}

/* OpenCV's HoughLinesStandard (the algorithm behind cv::HoughLines(image, lines, rho, theta,
 * threshold)): accumulator of (numangle+2) x (numrho+2) cells, votes for every non-zero pixel,
 * 4-neighbour local maxima above the threshold, sorted by votes. */
std::vector<std::pair<float, float>> RoadEstimation::HoughLines(const uint8_t* image, int rows,
                                                                int cols, float rho, float theta,
                                                                int threshold) {
    const int width = cols, height = rows;
    const float irho = 1 / rho;
    const double min_theta = 0, max_theta = 3.1415926535897932384626433832795;
    const int numangle = (int)lrint((max_theta - min_theta) / theta);
    const int numrho = (int)lrint(((width + height) * 2 + 1) / rho);
    std::vector<int> accum((size_t)(numangle + 2) * (numrho + 2), 0);
    std::vector<float> tabSin(numangle), tabCos(numangle);
    float ang = (float)min_theta;
    for (int n = 0; n < numangle; ang += theta, n++) {
        tabSin[n] = (float)(sin((double)ang) * irho);
        tabCos[n] = (float)(cos((double)ang) * irho);
    }
    for (int i = 0; i < height; i++)
        for (int j = 0; j < width; j++)
            if (image[(size_t)i * width + j] != 0)
                for (int n = 0; n < numangle; n++) {
                    int r = (int)lrint(j * tabCos[n] + i * tabSin[n]);
                    r += (numrho - 1) / 2;
                    accum[(size_t)(n + 1) * (numrho + 2) + r + 1]++;
                }
    std::vector<int> sort_buf;
    for (int r = 0; r < numrho; r++)
        for (int n = 0; n < numangle; n++) {
            const int base = (n + 1) * (numrho + 2) + r + 1;
            if (accum[base] > threshold && accum[base] > accum[base - 1] &&
                accum[base] >= accum[base + 1] && accum[base] > accum[base - numrho - 2] &&
                accum[base] >= accum[base + numrho + 2])
                sort_buf.push_back(base);
        }
    std::sort(sort_buf.begin(), sort_buf.end(), [&](int l1, int l2) {
        return accum[l1] > accum[l2] || (accum[l1] == accum[l2] && l1 < l2);
    });
    std::vector<std::pair<float, float>> lines;
    const double scale = 1. / (numrho + 2);
    for (int idx : sort_buf) {
        const int n = (int)floor(idx * scale) - 1;
        const int r = idx - (n + 1) * (numrho + 2) - 1;
        lines.emplace_back((r - (numrho - 1) * 0.5f) * rho, (float)min_theta + n * theta);
    }
    return lines;
}

bool RoadEstimation::ComputeHough(float& rho, float& theta, float& horizonPoint, float& pitch,
                                  float& cameraHeight, float& slope) { /* RE.cu:140-176 */
    /* (called from Compute, under its device guard) */
    IS_CHECK_RETURN(is_memcpy_d2h(m_vDisp.data(), d_vDispBinary, (size_t)m_max_dis * m_rows, m_stream));
    IS_CHECK_RETURN(is_stream_synchronize(m_stream));
    const auto lines = HoughLines(m_vDisp.data(), m_rows, m_max_dis, 1.0f, kPi / 180, m_HoughAccumThr);
    for (const auto& l : lines) {
        rho = std::abs(l.first);
        theta = l.second;
        ComputeCameraProperties(m_rows, rho, theta, horizonPoint, pitch, cameraHeight, slope);
        if (pitch >= m_minPitch && pitch <= m_maxPitch) return true;
    }
    return false;
}

void RoadEstimation::ComputeCameraProperties(int vdisp_rows, const float rho, const float theta,
                                             float& horizonPoint, float& pitch,
                                             float& cameraHeight, float& slope) const { /* :178-193 */
    horizonPoint = rho / sinf(theta);
    pitch = -atanf((m_cy - horizonPoint) / (m_focal)); /* y axis is inverted */
    const float last_row = (float)(vdisp_rows - 1);
    const float vDispDown = (rho - last_row * sinf(theta)) / cosf(theta);
    slope = (0 - vDispDown) / (horizonPoint - last_row);
    cameraHeight = m_b * cosf(pitch) / slope;
}
