/*
 * RoadEstimation.h -- road / camera-pose estimation from the v-disparity histogram, source-
 * compatible with /root/reference/InstanceStixels/include/InstanceStixels/RoadEstimation.h:32-94
 * (same public methods) but without OpenCV: the reference calls cv::HoughLines on the
 * binarised histogram (RoadEstimation.cu:153); here the standard Hough transform is implemented
 * in RoadEstimation.cpp following OpenCV's published HoughLinesStandard algorithm
 * (rho = 1, theta = pi/180, accumulator threshold 25, lines sorted by votes).
 */
#ifndef INSTANCESTIXELS_AMD_ROADESTIMATION_H_
#define INSTANCESTIXELS_AMD_ROADESTIMATION_H_

#include <stdint.h>

#include <vector>

#include "configuration.h"
#include "util.h"

class RoadEstimation {
public:
    RoadEstimation();
    ~RoadEstimation();

    void Initialize(const float camera_center_y, const float baseline, const float focal,
                    const int rows, const int cols, const int max_dis,
                    const float road_vdisparity_threshold = 0.2f);
    void Finish();

    bool Compute(const std::vector<pixel_t>& im);
    bool Compute(pixel_t* d_im);

    float GetCameraHeight() { return m_cameraHeight; }
    float GetPitch() { return m_pitch; }
    float GetSlope() { return m_slope; }
    int GetHorizonPoint() { return m_horizonPoint; }
    bool IsInitialized() { return m_is_initialized; }

    /* ---- additions (not in the reference) ----
     * Device the next Initialize() allocates on; default (-1): the calling thread's current HIP
     * device at Initialize() time.  Every later call (Compute, Finish) runs on that device
     * whatever the caller's current device is, on a stream of the object's own (an ordinary stream
     * that synchronises with the legacy NULL stream, like Stixels): together with
     * Stixels::SetDevice(d) the wrapper sequence GetInputDisparityImageOnDevice() ->
     * RoadEstimation::Compute(ptr) (apps/stixels_wrapper.cu:187) stays on device d. */
    void SetDevice(int device) { m_device = device; }
    int GetDevice() const { return m_device; }
    int GetActiveDevice() const { return m_ctx_device; }

    /* additions for tests */
    const std::vector<uint8_t>& GetBinaryVDisparity() const { return m_vDisp; }
    /* Standard Hough transform of a rows x cols 8-bit image; returns (rho, theta) pairs sorted
     * by accumulator votes (descending, ties by accumulator index). */
    static std::vector<std::pair<float, float>> HoughLines(const uint8_t* image, int rows, int cols,
                                                           float rho, float theta, int threshold);

private:
    void ComputeCameraProperties(int vdisp_rows, const float rho, const float theta,
                                 float& horizonPoint, float& pitch, float& cameraHeight,
                                 float& slope) const;
    bool ComputeHough(float& rho, float& theta, float& horizonPoint, float& pitch,
                      float& cameraHeight, float& slope);

    bool m_is_initialized = false;
    int m_device = -1;      /* requested (SetDevice) */
    int m_ctx_device = -1;  /* where the buffers of the last Initialize() live */
    void* m_stream = nullptr;
    pixel_t* d_disparity = nullptr;
    int* d_vDisp = nullptr;
    int* d_maximum = nullptr;
    uint8_t* d_vDispBinary = nullptr;
    std::vector<uint8_t> m_vDisp;

    int m_HoughAccumThr = 25;
    float m_binThr = 0.2f, m_maxPitch = 0, m_minPitch = 0;
    float m_maxCameraHeight = 0, m_minCameraHeight = 0;
    int m_max_dis = 0, m_rows = 0, m_cols = 0;
    float m_rho = 0, m_theta = 0;
    int m_horizonPoint = 0;
    float m_pitch = 0, m_cameraHeight = 0, m_cy = 0, m_b = 0, m_focal = 0, m_slope = 0;
};

#endif
