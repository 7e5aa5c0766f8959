/*
 * stixels_capi.cpp -- flat C view of the C++ `Stixels` host class, so that Python tests and
 * bench.py drive the SAME host code a C++ caller (run_cityscapes, StixelsWrapper) would:
 * SetConfig -> Initialize -> [SetDisparityImage, SetSegmentation, SetRoadParameters, Compute,
 * GetInstanceStixels] -> Finish  (call sequence of apps/run_cityscapes.cu:328-449).
 * C++ exceptions are turned into a negative return code + message.
 */
#include <cstring>
#include <ctime>
#include <exception>
#include <stdexcept>
#include <string>
#include <vector>

#include "instance_stixels_core.h"
#include "InstanceStixels/RoadEstimation.h"
#include "InstanceStixels/Stixels.hpp"

namespace {
thread_local std::string g_host_err;
template <class F>
int guard(F&& f) {
    try {
        f();
        return 0;
    } catch (const std::invalid_argument& e) {
        g_host_err = e.what();
        return -1;
    } catch (const std::exception& e) {
        g_host_err = e.what();
        return -2;
    }
}
}  // namespace

extern "C" {

/* C mirror of StixelConfig (types.h), field for field. */
struct ish_config {
    float rows, cols;
    int max_dis;
    float invalid_disparity, eps;
    int min_pts, size_filter, n_semantic_classes, n_offset_channels;
    float prior_weight, segmentation_weight, instance_weight, disparity_weight;
    int pairwise, column_step;
    float focal, baseline, camera_center_x, camera_center_y;
    float sigma_disparity_object, sigma_disparity_ground, sigma_sky;
    float pout, pout_sky, pord, pgrav, pblg;
    float pground_given_nexist, pobject_given_nexist, psky_given_nexist, pnexist_dis;
    float pground, pobject, psky;
    int width_margin;
    float sigma_camera_tilt, sigma_camera_height;
    int median_join;
    float epsilon, range_objects_z, road_vdisparity_threshold;
};

static StixelConfig to_cpp(const ish_config* c) {
    StixelConfig s;
    s.rows = c->rows; s.cols = c->cols; s.max_dis = c->max_dis;
    s.invalid_disparity = c->invalid_disparity; s.eps = c->eps; s.min_pts = c->min_pts;
    s.size_filter = c->size_filter; s.n_semantic_classes = c->n_semantic_classes;
    s.n_offset_channels = c->n_offset_channels; s.prior_weight = c->prior_weight;
    s.segmentation_weight = c->segmentation_weight; s.instance_weight = c->instance_weight;
    s.disparity_weight = c->disparity_weight; s.pairwise = c->pairwise != 0;
    s.column_step = c->column_step; s.focal = c->focal; s.baseline = c->baseline;
    s.camera_center_x = c->camera_center_x; s.camera_center_y = c->camera_center_y;
    s.sigma_disparity_object = c->sigma_disparity_object;
    s.sigma_disparity_ground = c->sigma_disparity_ground; s.sigma_sky = c->sigma_sky;
    s.pout = c->pout; s.pout_sky = c->pout_sky; s.pord = c->pord; s.pgrav = c->pgrav;
    s.pblg = c->pblg; s.pground_given_nexist = c->pground_given_nexist;
    s.pobject_given_nexist = c->pobject_given_nexist;
    s.psky_given_nexist = c->psky_given_nexist; s.pnexist_dis = c->pnexist_dis;
    s.pground = c->pground; s.pobject = c->pobject; s.psky = c->psky;
    s.width_margin = c->width_margin; s.sigma_camera_tilt = c->sigma_camera_tilt;
    s.sigma_camera_height = c->sigma_camera_height; s.median_join = c->median_join != 0;
    s.epsilon = c->epsilon; s.range_objects_z = c->range_objects_z;
    s.road_vdisparity_threshold = c->road_vdisparity_threshold;
    return s;
}

const char* ish_last_error(void) { return g_host_err.c_str(); }

void* ish_create(void) { return new Stixels(); }
void ish_destroy(void* h) { delete (Stixels*)h; }

int ish_set_config(void* h, const ish_config* c) {
    return guard([&] { ((Stixels*)h)->SetConfig(to_cpp(c)); });
}
int ish_initialize(void* h, int max_batch) {
    return guard([&] { ((Stixels*)h)->InitializeBatch(max_batch); });
}
int ish_precompute_host(void* h) {
    return guard([&] { ((Stixels*)h)->PrecomputeHost(); });
}
int ish_finish(void* h) {
    return guard([&] {
        if (((Stixels*)h)->IsInitialized()) ((Stixels*)h)->Finish();
    });
}
int ish_is_initialized(void* h) { return ((Stixels*)h)->IsInitialized() ? 1 : 0; }
int ish_real_cols(void* h) { return ((Stixels*)h)->GetRealCols(); }
int ish_max_sections(void* h) { return ((Stixels*)h)->GetMaxSections(); }

int ish_get_parameters(void* h, StixelParameters* out) {
    *out = ((Stixels*)h)->GetParameters();
    return 0;
}
int ish_get_luts(void* h, float* obj_cost_lut, float* obj_disparity_range) {
    Stixels* s = (Stixels*)h;
    std::memcpy(obj_cost_lut, s->GetObjectCostLUT().data(),
                s->GetObjectCostLUT().size() * sizeof(float));
    std::memcpy(obj_disparity_range, s->GetObjectDisparityRange().data(),
                s->GetObjectDisparityRange().size() * sizeof(float));
    return 0;
}
void* ish_core_context(void* h) { return ((Stixels*)h)->GetCoreContext(); }

int ish_set_disparity_image(void* h, const float* data, size_t n) {
    return guard([&] { ((Stixels*)h)->SetDisparityImage(std::vector<pixel_t>(data, data + n)); });
}
int ish_set_segmentation(void* h, const int32_t* data, size_t n) {
    return guard([&] { ((Stixels*)h)->SetSegmentation(std::vector<int32_t>(data, data + n)); });
}
int ish_set_road_parameters(void* h, int vhor, float tilt, float height, float alpha) {
    return guard([&] { ((Stixels*)h)->SetRoadParameters(vhor, tilt, height, alpha); });
}
int ish_get_ground_model(void* h, float* gf, float* ng, float* ig, int* vhor_lib) {
    return guard([&] {
        std::vector<float> a, b, c;
        ((Stixels*)h)->GetGroundModel(a, b, c, *vhor_lib);
        std::memcpy(gf, a.data(), a.size() * sizeof(float));
        std::memcpy(ng, b.data(), b.size() * sizeof(float));
        std::memcpy(ig, c.data(), c.size() * sizeof(float));
    });
}

/* Compute(): sections [realcols*max_sections]; header fields through `hdr[9]` =
 * rows, cols, realcols, max_sections, max_dis, column_step, semantic_classes, vhor, (unused). */
int ish_compute(void* h, int pairwise, Section* sections, int* hdr, float* alpha_ground,
                float* ret) {
    return guard([&] {
        StixelsData d;
        *ret = ((Stixels*)h)->Compute(pairwise != 0, d);
        std::memcpy(sections, d.sections.data(), d.sections.size() * sizeof(Section));
        hdr[0] = d.rows; hdr[1] = d.cols; hdr[2] = d.realcols; hdr[3] = d.max_sections;
        hdr[4] = d.max_dis; hdr[5] = d.column_step; hdr[6] = d.semantic_classes; hdr[7] = d.vhor;
        *alpha_ground = d.alpha_ground;
    });
}

/* Times n_iter calls of Stixels::Compute() on the frame set before (SetDisparityImage /
 * SetSegmentation / SetRoadParameters), StixelsData reused like a C++ caller's loop would; with
 * `with_instances` every frame also fetches GetInstanceStixels().  -> seconds per frame. */
int ish_time_compute(void* h, int pairwise, int n_iter, int with_instances, double* s_per_frame) {
    return guard([&] {
        Stixels* s = (Stixels*)h;
        StixelsData d;
        for (int i = 0; i < 3; i++) s->Compute(pairwise != 0, d);
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        size_t sink = 0;
        for (int i = 0; i < n_iter; i++) {
            s->Compute(pairwise != 0, d);
            if (with_instances) sink += s->GetInstanceStixels().size();
        }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        (void)sink;
        *s_per_frame = ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec)) / n_iter;
    });
}

/* ComputeBatch() on device-resident inputs.  road: [n][4] = (vhor_image, camera_tilt,
 * camera_height, alpha_ground); sections: [n][realcols*max_sections]; vhor_lib: [n].
 * triples (optional): [n][cap][3] (column, section, label) of every frame, counts: [n]. */
int ish_compute_batch(void* h, int pairwise, int n_images, const float* d_big, const int32_t* d_seg,
                      const float* road, Section* sections, int* vhor_lib, int* triples, int cap,
                      int* counts, void* stream) {
    return guard([&] {
        Stixels* s = (Stixels*)h;
        std::vector<Stixels::RoadParameters> rp(n_images);
        for (int i = 0; i < n_images; i++)
            rp[i] = Stixels::RoadParameters{(int)road[4 * i], road[4 * i + 1], road[4 * i + 2], road[4 * i + 3]};
        std::vector<StixelsData> out;
        std::vector<Stixels::InstanceMapping> maps;
        s->ComputeBatch(pairwise != 0, n_images, d_big, d_seg, rp.data(), out, stream,
                        triples ? &maps : nullptr);
        for (int i = 0; i < n_images; i++) {
            std::memcpy(sections + (size_t)i * out[i].sections.size(), out[i].sections.data(),
                        out[i].sections.size() * sizeof(Section));
            vhor_lib[i] = out[i].vhor;
            if (triples) {
                int n = 0;
                for (const auto& kv : maps[i]) {
                    if (n >= cap) break;
                    int* t = triples + ((size_t)i * cap + n) * 3;
                    t[0] = kv.first.first; t[1] = kv.first.second; t[2] = kv.second;
                    n++;
                }
                counts[i] = (int)maps[i].size();
            }
        }
    });
}

/* ComputeBatchGather(): this rank's shard + the RCCL gather of every rank's Sections on `dst`.
 * road: [n][4] of this rank; images_per_rank: [ranks]; road_all: [sum][4] (dst only, else null);
 * sections_all: [sum][realcols*max_sections] on dst; *n_out: frames written (0 on the other ranks). */
int ish_compute_batch_gather(void* h, int pairwise, int n_images, const float* d_big, const int32_t* d_seg,
                             const float* road, void* comm, int dst, const int* images_per_rank,
                             const float* road_all, int n_all, Section* sections_all, int* vhor_all,
                             int* n_out, void* stream) {
    return guard([&] {
        Stixels* s = (Stixels*)h;
        auto conv = [](const float* r, int n) {
            std::vector<Stixels::RoadParameters> rp(n);
            for (int i = 0; i < n; i++)
                rp[i] = Stixels::RoadParameters{(int)r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]};
            return rp;
        };
        if (road_all) { /* dst: the caller sized road_all / sections_all / vhor_all for n_all frames */
            int rank = 0, nranks = 0;
            if (is_comm_rank(comm, &rank, &nranks) != IS_OK)
                throw std::runtime_error(std::string("ish_compute_batch_gather: ") + is_last_error());
            long sum = 0;
            for (int r = 0; r < nranks; r++) sum += images_per_rank[r];
            if (sum != (long)n_all)
                throw std::invalid_argument("ish_compute_batch_gather: n_all differs from the sum of images_per_rank");
        }
        const std::vector<Stixels::RoadParameters> rp = conv(road, n_images);
        const std::vector<Stixels::RoadParameters> ra = road_all ? conv(road_all, n_all)
                                                                 : std::vector<Stixels::RoadParameters>();
        std::vector<StixelsData> out;
        s->ComputeBatchGather(pairwise != 0, n_images, d_big, d_seg, rp.data(), comm, dst, images_per_rank,
                              road_all ? ra.data() : nullptr, out, stream);
        *n_out = (int)out.size();
        for (size_t i = 0; i < out.size() && i < (size_t)(n_all > 0 ? n_all : 0); i++) {
            std::memcpy(sections_all + i * out[i].sections.size(), out[i].sections.data(),
                        out[i].sections.size() * sizeof(Section));
            vhor_all[i] = out[i].vhor;
        }
    });
}

/* Times n_iter ComputeBatch() calls of n_images frames (inputs resident on the device), with or
 * without the per-frame instance mappings.  -> seconds per call. */
int ish_time_compute_batch(void* h, int pairwise, int n_images, const float* d_big, const int32_t* d_seg,
                           const float* road, int n_iter, int with_instances, double* s_per_call) {
    return guard([&] {
        Stixels* s = (Stixels*)h;
        std::vector<Stixels::RoadParameters> rp(n_images);
        for (int i = 0; i < n_images; i++)
            rp[i] = Stixels::RoadParameters{(int)road[4 * i], road[4 * i + 1], road[4 * i + 2], road[4 * i + 3]};
        std::vector<StixelsData> out;
        std::vector<Stixels::InstanceMapping> maps;
        s->ComputeBatch(pairwise != 0, n_images, d_big, d_seg, rp.data(), out, nullptr,
                        with_instances ? &maps : nullptr);
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int i = 0; i < n_iter; i++)
            s->ComputeBatch(pairwise != 0, n_images, d_big, d_seg, rp.data(), out, nullptr,
                            with_instances ? &maps : nullptr);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        *s_per_call = ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec)) / n_iter;
    });
}

int ish_set_device(void* h, int device) {
    return guard([&] { ((Stixels*)h)->SetDevice(device); });
}

/* GetInstanceStixels(): triples (column, section, label); returns the count (<= cap) or <0. */
int ish_get_instance_stixels(void* h, int* triples, int cap) {
    int n = 0;
    const int rc = guard([&] {
        const auto m = ((Stixels*)h)->GetInstanceStixels();
        for (const auto& kv : m) {
            if (n >= cap) break;
            triples[3 * n] = kv.first.first;
            triples[3 * n + 1] = kv.first.second;
            triples[3 * n + 2] = kv.second;
            n++;
        }
    });
    return rc ? rc : n;
}

int ish_get_3d_vertices(void* h, const Section* sections, float alpha_ground, int vhor,
                        float* out, int cap) {
    int n = 0;
    const int rc = guard([&] {
        Stixels* s = (Stixels*)h;
        StixelsData d;
        d.sections.assign(sections, sections + (size_t)s->GetRealCols() * s->GetMaxSections());
        d.alpha_ground = alpha_ground;
        d.vhor = vhor;
        const std::vector<float> v = s->Get3DVertices(d);
        n = (int)std::min<size_t>(v.size(), (size_t)cap);
        std::memcpy(out, v.data(), (size_t)n * sizeof(float));
    });
    return rc ? rc : n;
}

int ish_save_stixels(void* h, Section* sections, const int* triples, int n_triples,
                     float alpha_ground, int vhor, const char* fname) {
    return guard([&] {
        Stixels* s = (Stixels*)h;
        std::map<std::pair<int, int>, int> m;
        for (int i = 0; i < n_triples; i++)
            m[std::make_pair(triples[3 * i], triples[3 * i + 1])] = triples[3 * i + 2];
        Stixels::SaveStixels(sections, m, alpha_ground, vhor, s->GetRealCols(),
                             s->GetMaxSections(), fname);
    });
}

/* ---- RoadEstimation (f3) ---- */
void* ire_create(void) { return new RoadEstimation(); }
void ire_destroy(void* h) { delete (RoadEstimation*)h; }
int ire_initialize(void* h, float cy, float baseline, float focal, int rows, int cols, int max_dis,
                   float threshold) {
    return guard([&] { ((RoadEstimation*)h)->Initialize(cy, baseline, focal, rows, cols, max_dis, threshold); });
}
int ire_finish(void* h) {
    return guard([&] {
        if (((RoadEstimation*)h)->IsInitialized()) ((RoadEstimation*)h)->Finish();
    });
}
/* returns 1 if a road line was found; out = pitch, camera height, slope, horizon point */
int ire_compute(void* h, const float* image, size_t n, float* out4) {
    int ok = 0;
    const int rc = guard([&] {
        RoadEstimation* r = (RoadEstimation*)h;
        ok = r->Compute(std::vector<pixel_t>(image, image + n)) ? 1 : 0;
        out4[0] = r->GetPitch(); out4[1] = r->GetCameraHeight(); out4[2] = r->GetSlope();
        out4[3] = (float)r->GetHorizonPoint();
    });
    return rc ? rc : ok;
}
int ire_set_device(void* h, int device) {
    return guard([&] { ((RoadEstimation*)h)->SetDevice(device); });
}
int ire_active_device(void* h) { return ((RoadEstimation*)h)->GetActiveDevice(); }
/* Compute(pixel_t* d_im): the disparity image is on the device already -- what the wrapper passes
 * after Stixels::GetInputDisparityImageOnDevice() (apps/stixels_wrapper.cu:187) */
int ire_compute_device(void* h, float* d_image, float* out4) {
    int ok = 0;
    const int rc = guard([&] {
        RoadEstimation* r = (RoadEstimation*)h;
        ok = r->Compute(d_image) ? 1 : 0;
        out4[0] = r->GetPitch(); out4[1] = r->GetCameraHeight(); out4[2] = r->GetSlope();
        out4[3] = (float)r->GetHorizonPoint();
    });
    return rc ? rc : ok;
}
void* ish_get_input_disparity_on_device(void* h) {
    return (void*)((Stixels*)h)->GetInputDisparityImageOnDevice();
}
int ire_get_binary(void* h, uint8_t* out, size_t n) {
    const auto& v = ((RoadEstimation*)h)->GetBinaryVDisparity();
    std::memcpy(out, v.data(), std::min(n, v.size()));
    return 0;
}
int ire_hough_lines(const uint8_t* image, int rows, int cols, float rho, float theta, int threshold,
                    float* out, int cap) {
    const auto lines = RoadEstimation::HoughLines(image, rows, cols, rho, theta, threshold);
    const int n = (int)std::min<size_t>(lines.size(), (size_t)cap);
    for (int i = 0; i < n; i++) { out[2 * i] = lines[i].first; out[2 * i + 1] = lines[i].second; }
    return (int)lines.size();
}

} /* extern "C" */
