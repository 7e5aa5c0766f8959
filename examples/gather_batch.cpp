/*
 * gather_batch.cpp -- a plain C++ caller (g++, no HIP header, no Python) of the multi-GPU path: every rank runs its
 * shard of a batch through Stixels::ComputeBatchGather and rank 0 receives the Sections of all ranks over RCCL
 * (INTEGRATION.md section 4).  The reference has no counterpart: apps/run_cityscapes.cu:245-449 walks its frames one
 * by one on one GPU.  Started as ONE process this is a one-rank communicator (what the GPU test runs: the RCCL
 * calls, the go-ahead protocol and the unpack are the same code); with N processes pass
 *     gather_batch <rank> <nranks> <id file> [device]   (rank 0 writes the 128-byte communicator id to the file
 * first; device defaults to the rank).  Rank 0 compares what it received with ComputeBatch of the same frames --
 * its own and, regenerated from their seeds, those of every other rank.
 *
 *   g++ -std=c++17 -O2 -Iinclude -Iinclude/InstanceStixels examples/gather_batch.cpp \
 *       -Linstance_stixels_amd/lib -lInstanceStixels -lis_core -Wl,-rpath,$PWD/instance_stixels_amd/lib \
 *       -o examples/gather_batch
 */
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "Stixels.hpp"
#include "instance_stixels_core.h"

#define CHECK(x)                                                                          \
    do {                                                                                  \
        if ((x) != 0) { fprintf(stderr, "%s failed: %s\n", #x, is_last_error()); return 1; } \
    } while (0)

int main(int argc, char** argv) {
    const int rank = argc > 1 ? atoi(argv[1]) : 0;
    const int nranks = argc > 2 ? atoi(argv[2]) : 1;
    const char* id_file = argc > 3 ? argv[3] : nullptr;
    const int device = argc > 4 ? atoi(argv[4]) : rank;
    const int rows = 128, cols = 256, max_dis = 32, frames = 3; /* per rank */

    /* ---- the communicator: rank 0 creates the id, the others read it */
    char id[128];
    if (rank == 0) {
        CHECK(is_comm_unique_id(id, sizeof id));
        if (id_file) { /* (written under another name and renamed: a reader never sees half an id) */
            const std::string tmp = std::string(id_file) + ".tmp";
            FILE* f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) return 1;
            fclose(f);
            if (rename(tmp.c_str(), id_file) != 0) return 1;
        }
    } else {
        FILE* f = nullptr;
        while (!(f = fopen(id_file, "rb"))) {}
        if (fread(id, 1, sizeof id, f) != sizeof id) return 1;
        fclose(f);
    }
    CHECK(is_set_device(device));
    void* comm = nullptr;
    CHECK(is_comm_init_rank(&comm, nranks, id, rank));

    StixelConfig cfg;
    cfg.rows = rows; cfg.cols = cols; cfg.max_dis = max_dis; cfg.column_step = 8;
    cfg.invalid_disparity = -1.0f; cfg.n_semantic_classes = 19; cfg.n_offset_channels = 2;
    cfg.prior_weight = 1e4f; cfg.segmentation_weight = 11.241965f; cfg.instance_weight = 0.001731f;
    cfg.disparity_weight = 0.006993f; cfg.eps = 23.89408f; cfg.min_pts = 4; cfg.size_filter = 42;
    cfg.focal = 2262.52f; cfg.baseline = 0.209313f; cfg.camera_center_x = 0.5f * cols; cfg.camera_center_y = 0.5f * rows;

    /* ---- this rank's frames (seeded by the rank), resident on its device */
    const int vhor_img = (int)(0.45f * rows), realcols = cols / 8;
    const float alpha = 0.8f * max_dis / (rows - vhor_img);
    const int p2s = (int)powf(2, ceilf(log2f(rows / 8 + 1)));
    std::vector<float> disp((size_t)frames * rows * cols);
    std::vector<int32_t> seg((size_t)frames * realcols * 21 * p2s, 0);
    auto generate = [&](int of_rank) {
        std::mt19937 rng(100 + of_rank);
        std::uniform_real_distribution<float> U(0.0f, 1.0f);
        for (int f = 0; f < frames; f++) {
            for (int r = 0; r < rows; r++)
                for (int c = 0; c < cols; c++) {
                    const float d = r > vhor_img ? alpha * (r - vhor_img) + U(rng) : 0.5f * U(rng);
                    disp[((size_t)f * rows + r) * cols + c] = std::fmin(std::fmax(d, 0.01f), max_dis - 1.01f);
                }
            for (int c = 0; c < realcols; c++)
                for (int ch = 0; ch < 19; ch++)
                    for (int k = 0; k < rows / 8; k++) {
                        const int truth = (rows - 1 - (8 * k + 4)) > vhor_img ? 0 : 10;
                        seg[(((size_t)f * realcols + c) * 21 + ch) * p2s + k] = (ch == truth ? 1 : 30) + (int)(4 * U(rng));
                    }
        }
    };
    generate(rank);
    float* d_disp = nullptr;
    int32_t* d_seg = nullptr;
    CHECK(is_device_malloc((void**)&d_disp, disp.size() * sizeof(float)));
    CHECK(is_device_malloc((void**)&d_seg, seg.size() * sizeof(int32_t)));
    CHECK(is_memcpy_h2d(d_disp, disp.data(), disp.size() * sizeof(float), nullptr));
    CHECK(is_memcpy_h2d(d_seg, seg.data(), seg.size() * sizeof(int32_t), nullptr));
    CHECK(is_device_synchronize());

    Stixels st;
    st.SetConfig(cfg);
    st.SetDevice(device);
    st.InitializeBatch(frames);
    std::vector<Stixels::RoadParameters> road(frames, Stixels::RoadParameters{vhor_img, 0.05f, 1.2f, alpha});
    std::vector<Stixels::RoadParameters> road_all((size_t)frames * nranks, road[0]);
    std::vector<int> images_per_rank(nranks, frames);

    std::vector<StixelsData> mine, all;
    st.ComputeBatch(false, frames, d_disp, d_seg, road.data(), mine);
    st.ComputeBatchGather(false, frames, d_disp, d_seg, road.data(), comm, /*dst=*/0, images_per_rank.data(),
                          rank == 0 ? road_all.data() : nullptr, all);
    int bad = 0;
    if (rank == 0) {
        if ((int)all.size() != frames * nranks) bad++;
        for (int r = 0; r < nranks && !bad; r++) { /* the frames arrive in rank order */
            if (r > 0) { /* rank r's frames again, from its seed, through ComputeBatch here */
                generate(r);
                CHECK(is_memcpy_h2d(d_disp, disp.data(), disp.size() * sizeof(float), nullptr));
                CHECK(is_memcpy_h2d(d_seg, seg.data(), seg.size() * sizeof(int32_t), nullptr));
                CHECK(is_device_synchronize());
                st.ComputeBatch(false, frames, d_disp, d_seg, road.data(), mine);
            }
            for (int f = 0; f < frames && !bad; f++)
                for (int c = 0; c < realcols; c++)
                    for (int i = 0; i < st.GetMaxSections(); i++) {
                        const Section& a = mine[f].sections[(size_t)c * st.GetMaxSections() + i];
                        const Section& b = all[(size_t)r * frames + f].sections[(size_t)c * st.GetMaxSections() + i];
                        if (a.type != b.type) { bad++; break; }
                        if (a.type == -1) break; /* entries behind the terminator are unspecified */
                        if (memcmp(&a, &b, sizeof(Section)) != 0) { bad++; break; }
                    }
        }
        size_t n = 0;
        for (const StixelsData& d : all)
            for (int c = 0; c < d.realcols; c++)
                for (int i = 0; i < d.max_sections && d.sections[(size_t)c * d.max_sections + i].type != -1; i++) n++;
        printf("rank 0 holds %zu frames of %d ranks, %zu stixels; gathered == computed: %s\n", all.size(), nranks, n,
               bad ? "NO" : "yes");
    }
    st.Finish();
    CHECK(is_device_free(d_disp));
    CHECK(is_device_free(d_seg));
    CHECK(is_comm_destroy(comm));
    return bad ? 1 : 0;
}
