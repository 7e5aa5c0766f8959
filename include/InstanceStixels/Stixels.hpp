/*
 * Stixels.hpp -- host class of the MI355X-native stixel library.
 *
 * Public surface = the reference's `class Stixels`
 * (/root/reference/InstanceStixels/include/InstanceStixels/Stixels.hpp:40-96): same method
 * names, argument meaning and error behaviour, so run_cityscapes / StixelsWrapper / the ROS node
 * compile against it unchanged.  Differences, all behind the same API:
 *   - plain C++ header (no CUDA/HIP types): callers need not be device translation units;
 *   - the device side is the C ABI of include/instance_stixels_core.h (HIP, gfx950);
 *   - Compute() does not modify the segmentation buffer it is given;
 *   - ComputeBatch() / InitializeBatch() are additions for batched, multi-GPU use.
 */
#ifndef INSTANCESTIXELS_AMD_STIXELS_HPP_
#define INSTANCESTIXELS_AMD_STIXELS_HPP_

#include <stdint.h>

#include <cmath>
#include <map>
#include <utility>
#include <vector>

#include "configuration.h"
#include "types.h"
#include "util.h"

constexpr float PIFLOAT = 3.1416f;

class Stixels {
public:
    Stixels();
    ~Stixels();

    void Initialize();
    void Finish();

    float Compute(bool pairwise, StixelsData& stixels, int32_t* d_segmentation_local = nullptr);
    float ClusterInstances();
    std::map<std::pair<int, int>, int> GetInstanceStixels();
    int GetRealCols();
    int GetMaxSections();
    void SetConfig(const StixelConfig& config);
    void SetSegmentation(const std::vector<int32_t>& segmentation);
    void SetSegmentationParameters(const int classes, const int instance_channels);
    void SetClusteringParameters(const float eps, const int min_pts, const int size_filter);
    void SetWeightParameters(const float prior_weight, const float disparity_weight,
                             const float segmentation_weight, const float instance_weight);
    void SetDisparityImage(const std::vector<pixel_t>& disp_im);
    pixel_t* GetInputDisparityImageOnDevice();
    void SetProbabilities(float pout, float pout_sky, float pground_given_nexist,
                          float pobject_given_nexist, float psky_given_nexist, float pnexist_dis,
                          float pground, float pobject, float psky, float pord, float pgrav,
                          float pblg);
    void SetRoadParameters(int vhor, float camera_tilt, float camera_height, float alpha_ground);
    void SetCameraParameters(float focal, float baseline, float sigma_camera_tilt,
                             float sigma_camera_height, float camera_center_x = -1,
                             float camera_center_y = -1);
    void SetDisparityParameters(const int rows, const int cols, const int max_dis,
                                const float invalid_disparity, const float sigma_disparity_object,
                                const float sigma_disparity_ground, float sigma_sky);
    void SetModelParameters(const int column_step, const bool median_join, float epsilon,
                            float range_objects_z, int width_margin);
    std::vector<float> Get3DVertices(const StixelsData& stixels_data);
    static void SaveStixels(Section* stixels, std::map<std::pair<int, int>, int> instance_stixels,
                            const float alpha_ground, const int vhor, const int real_cols,
                            const int max_segments, const char* fname);
    bool IsInitialized() { return m_is_initialized; }

    /* ---- additions (not in the reference) ---- */
    /* Road parameters of one frame of a batch, image-convention vhor as in SetRoadParameters. */
    struct RoadParameters {
        int vhor;
        float camera_tilt, camera_height, alpha_ground;
    };
    /* Device the next Initialize() creates its buffers on; default (-1): the calling thread's
     * current HIP device at Initialize() time (one process per GPU: select the device before, as
     * with the reference's CUDA code).  Every later call runs on that device whatever the
     * caller's current device is. */
    void SetDevice(int device) { m_device = device; }
    /* the device requested with SetDevice() (-1 = current device at Initialize() time) */
    int GetDevice() const { return m_device; }
    /* the device the buffers of the last Initialize() live on (-1 before) */
    int GetActiveDevice() const { return m_ctx_device; }
    /* Host half of Initialize() (tables + parameter block); needs no device. */
    void PrecomputeHost();
    /* Like Initialize(), with device scratch for up to max_batch frames per ComputeBatch(). */
    void InitializeBatch(int max_batch);
    /* Batched Compute on device-resident inputs:
     *   d_disparity_big [n][rows][cols] float, d_segmentation [n][realcols][channels][P2S] int32.
     * Fills `out[i]` like Compute() fills its StixelsData.  `stream` is a hipStream_t.
     * `instance_stixels` (optional): filled with what GetInstanceStixels() returns after a
     * Compute() of frame i -- the instance candidates of every frame are compacted and clustered
     * on the device in two launches for the whole batch (the reference does both inside
     * Compute(), StixelsKernels.cu:926-942, Stixels.cu:613). */
    typedef std::map<std::pair<int, int>, int> InstanceMapping;
    void ComputeBatch(bool pairwise, int n_images, const pixel_t* d_disparity_big,
                      const int32_t* d_segmentation, const RoadParameters* road,
                      std::vector<StixelsData>& out, void* stream = nullptr,
                      std::vector<InstanceMapping>* instance_stixels = nullptr);
    /* Multi-GPU (an addition: the reference runs on one GPU): this rank's shard of a batch through
     * ComputeBatch's device path, then the compacted final gather of EVERY rank's Sections on rank `dst` of
     * `comm` over RCCL (an ncclComm_t passed as void*; plain-C++ callers create it with is_comm_unique_id /
     * is_comm_init_rank of instance_stixels_core.h).  One process per GPU, every rank calls it.
     *   images_per_rank  [ranks]: the shard sizes, known to every rank; n_images = its own entry
     *   road_all         on dst: the road parameters of ALL frames in rank order (dst hands out the work,
     *                    so it has them) for the StixelsData headers; ignored elsewhere
     * On dst `out` holds the frames of all ranks in rank order, elsewhere it is left empty.  Sections
     * only: the instance mappings stay with the rank that computed them (ComputeBatch). */
    void ComputeBatchGather(bool pairwise, int n_images, const pixel_t* d_disparity_big,
                            const int32_t* d_segmentation, const RoadParameters* road, void* comm, int dst,
                            const int* images_per_rank, const RoadParameters* road_all,
                            std::vector<StixelsData>& out, void* stream = nullptr);
    /* Introspection for tests / bench. */
    const StixelParameters& GetParameters() const { return m_params; }
    const std::vector<float>& GetObjectCostLUT() const { return m_obj_cost_lut; }
    const std::vector<float>& GetObjectDisparityRange() const { return m_object_disparity_range; }
    void GetGroundModel(std::vector<float>& ground_function,
                        std::vector<float>& normalization_ground,
                        std::vector<float>& inv_sigma2_ground, int& vhor_lib);
    is_ctx* GetCoreContext() { return m_ctx; }

private:
    struct GroundModel {
        std::vector<float> function, normalization, inv_sigma2;
    };
    void PrecomputeSky();
    void PrecomputeObject();
    void PrecomputeGround(int vhor_lib, float camera_tilt, float camera_height, float alpha_ground,
                          GroundModel& out) const;
    float GetDataCostObject(const int fn, const int dis) const;
    float ComputeObjectDisparityRange(const float previous_mean) const;
    float FastLog(float v) const;
    void FillHeader(StixelsData& d, float alpha_ground, int vhor_lib) const;
    void EnsurePackBuffers();
    is_instance_buffers InstanceBuffers(int image = 0) const;
    GroundModel m_ground; /* per-frame ground model, storage reused between frames */
    /* the road parameters m_ground was computed for: a frame with the same parameters (a fixed
     * camera model, a replayed sequence) reuses it -- 1024 rows of erf / sqrt / log on the host
     * are 12 us of a 0.26 ms frame.  Invalidated by Initialize(). */
    float m_ground_key[12] = {0};  /* every input of PrecomputeGround */
    bool m_ground_valid = false;

    /* device (owned between Initialize and Finish, Stixels.cu:53-74, 136-163) */
    is_ctx* m_ctx = nullptr;
    pixel_t* d_disparity = nullptr;
    pixel_t* d_disparity_big = nullptr;
    int32_t* d_segmentation = nullptr;
    Section* d_stixels = nullptr;
    float* d_instance_centerofmass = nullptr;
    int32_t* d_instance_indices = nullptr;
    uint8_t* d_instance_core_candidates = nullptr;
    int32_t* d_instances_per_class = nullptr;
    /* (every instance array: one slice per frame of the batch) */
    int32_t* d_instance_labels = nullptr;  /* the reference's d_instance_labels, Stixels.cu:66-68 */
    int32_t* d_instance_packed = nullptr;  /* [1 + 3*classes*realcols*max_sections], see is_instance_buffers */
    /* pinned host mirrors: Compute() ends with ONE stream synchronisation */
    Section* h_stixels = nullptr;
    /* One device block: header rows (the per-class candidate counts, d_instances_per_class) in
     * front of the sections (d_stixels).  Compute() fetches the header and the first
     * m_head_sections sections of every column with ONE pitched copy into h_stixels_head; a column
     * without a terminator among them (rare) makes it fetch the complete array. */
    Section* d_stixels_block = nullptr;
    Section* h_stixels_head = nullptr;
    int m_header_rows = 0;
    int m_head_sections = 0;
    int32_t* h_instance_head = nullptr;    /* [max_batch][8 per-class counts] */
    /* ComputeBatchGather: the packed payload of this rank and, on the destination, the landing buffers
     * (allocated on first use, grown on demand, released by Finish) */
    int32_t* d_pack_counts = nullptr;
    int32_t* d_pack_offsets = nullptr;
    Section* d_pack_sections = nullptr;
    int32_t* d_all_counts = nullptr;
    Section* d_all_packed = nullptr;
    Section* d_all_sections = nullptr;
    size_t m_all_columns_cap = 0, m_all_packed_cap = 0;
    /* ComputeBatch: the same packed payload copied to the host in two pinned pieces (offsets, used sections) */
    int32_t* h_pack_offsets = nullptr;
    Section* h_pack_sections = nullptr;
    size_t m_h_pack_cap = 0; /* sections h_pack_sections holds */
    int32_t* h_all_counts = nullptr; /* ComputeBatchGather on dst: the per-column counts of all ranks (pinned) */
    size_t m_h_all_counts_cap = 0;
    int32_t* h_instance_packed = nullptr;
    /* every device operation of the object runs on this stream (an ordinary stream: it still
     * synchronises with work the caller queued on the legacy NULL stream, like the reference's
     * default-stream code; on the NULL stream itself the auxiliary streams of the core never
     * overlap) */
    void* m_stream = nullptr;
    int m_max_batch = 1;
    int m_device = -1;      /* requested (SetDevice) */
    int m_ctx_device = -1;  /* resolved at Initialize(): where the buffers live */
    bool m_labels_on_host = false; /* h_instance_packed holds the triples of the last frame */

    StixelParameters m_params{};
    int m_max_sections = MAX_STIXELS_PER_COLUMN;
    bool m_is_initialized = false;

    /* probabilities */
    float m_pout = 0, m_pout_sky = 0;
    float m_pnexists_given_ground = 0, m_pnexists_given_object = 0, m_pnexists_given_sky = 0;
    float m_pord = 0, m_pgrav = 0, m_pblg = 0;
    /* camera */
    float m_focal = 0, m_baseline = 0, m_camera_tilt = 0, m_sigma_camera_tilt = 0;
    float m_camera_height = 0, m_sigma_camera_height = 0;
    float m_camera_center_x = -1, m_camera_center_y = -1;
    int m_vhor = 0;
    /* segmentation / weights */
    int m_segmentation_classes = 0, m_segmentation_channels = 0;
    float m_prior_weight = 0, m_disparity_weight = 0, m_segmentation_weight = 0,
          m_instance_weight = 0;
    /* disparity */
    int m_max_dis = 0;
    float m_max_disf = 0, m_invalid_disparity = -1;
    int m_rows = 0, m_cols = 0, m_realcols = 0;
    float m_sigma_disparity_object = 0, m_sigma_disparity_ground = 0, m_sigma_sky = 0;
    /* model */
    int m_column_step = 0;
    bool m_median_join = false;
    float m_alpha_ground = 0, m_range_objects_z = 0, m_epsilon = 0;
    int m_width_margin = 0;
    /* tables */
    std::vector<float> m_log_lut, m_obj_cost_lut, m_object_disparity_range;
    std::vector<float> m_normalization_object, m_inv_sigma2_object;
    float m_max_dis_log = 0, m_rows_log = 0;
    float m_puniform = 0, m_puniform_sky = 0, m_normalization_sky = 0, m_inv_sigma2_sky = 0;
    /* instances (host mirrors) */
    std::vector<int> m_instances_per_class;
    int m_instance_classes = IS_INSTANCE_CLASSES;
};

#endif
