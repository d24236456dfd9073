// The reference-shaped call, timed end to end (round 5, VERDICT r4 "next round" 3): what a C++ caller of the reference
// actually invokes is CollisionMapGrid::ExtractSignedDistanceField(oob_value, unknown_is_filled, add_virtual_border)
// (reference include/sdf_tools/collision_map.hpp:680-712; the pybind surface calls the same method, src/sdf_tools/
// bindings.cpp:81) -- a host container of 8-byte COLLISION_CELL records in, a host SignedDistanceField out.  This client
// builds an n^3 map of random occupancy, times the method the way such a caller sees it (construction of the result,
// upload, kernels, download, return by move; the caller's later destruction of the result is its own cost and is kept out
// of the timed region), checks the field bit for bit against the raw C ABI (sdfgpu_build on the mask the same predicate
// yields, host_pack off: the upload-and-classify-on-device path of rounds 1 - 4) and prints one JSON line.
//
//   class_seam [n_cells = 256] [reps = 3] [p_filled = 0.5] [--no-gpu]
// Build: see tests/test_cpp_example.py (g++ -O2 -std=c++17 -I include ... -lsdfgpu -lz).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "sdf_tools/collision_map.hpp"
#include "sdf_tools/sdf.hpp"

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
    int64_t n = 256;
    int reps = 3;
    double p = 0.5;
    bool no_gpu = false;
    int pos = 0;
    for (int i = 1; i < argc; ++i) {
        if (std::string(argv[i]) == "--no-gpu") no_gpu = true;
        else if (pos == 0) { n = std::atoll(argv[i]); ++pos; }
        else if (pos == 1) { reps = std::atoi(argv[i]); ++pos; }
        else { p = std::atof(argv[i]); ++pos; }
    }
    const double resolution = 0.01;
    sdf_tools::CollisionMapGrid map(Eigen::Isometry3d::Identity(), "world", resolution, n, n, n, sdf_tools::COLLISION_CELL(0.0));
    if (no_gpu) {
        try {
            map.ExtractSignedDistanceField(INFINITY, false, false);
        } catch (const std::runtime_error& e) {
            std::printf("no GPU: %s\n", e.what());               // expected on a CPU-only box: no CPU fallback
            return 0;
        }
        std::printf("a GPU is present\n");
        return 0;
    }
    // random occupancy: filled 1.0 with probability p, unknown (0.5) on 1 % of the cells, free otherwise -- the unknown cells
    // exercise the second half of the predicate (collision_map.hpp:689-704) with unknown_is_filled = true below
    std::mt19937_64 rng(5);
    std::vector<sdf_tools::COLLISION_CELL>& cells = map.GetMutableRawData();
    const size_t N = cells.size();
    for (size_t i = 0; i < N; ++i) {
        const double u = (double)(rng() >> 11) * (1.0 / 9007199254740992.0);
        cells[i] = sdf_tools::COLLISION_CELL(u < p ? 1.0f : (u < p + 0.01 ? 0.5f : 0.0f));
    }
    const bool unknown_is_filled = true;

    map.ExtractSignedDistanceField(INFINITY, unknown_is_filled, false);           // warm-up: context, device buffers, pinned staging
    std::vector<double> ms;
    std::pair<sdf_tools::SignedDistanceField, std::pair<double, double>> last;
    for (int r = 0; r < reps; ++r) {
        const double t0 = now_ms();
        auto res = map.ExtractSignedDistanceField(INFINITY, unknown_is_filled, false);
        ms.push_back(now_ms() - t0);
        last = std::move(res);                                                     // (the previous result is destroyed here, untimed)
        if (!res.first.GetImmutableRawData().empty()) { std::printf("FAIL: moving a SignedDistanceField copied its array\n"); return 1; }
    }
    // the same field through the raw C ABI, with the classification done on the device from an uploaded mask
    std::vector<uint8_t> mask(N);
    for (size_t i = 0; i < N; ++i) mask[i] = (cells[i].occupancy > 0.5f || (unknown_is_filled && cells[i].occupancy == 0.5f)) ? 1 : 0;
    std::vector<float> ref(N);
    double rmax = 0.0, rmin = 0.0;
    {
        const std::shared_ptr<sdf_generation::SharedGpuContext> ctx = sdf_generation::GpuContext::Shared();
        const std::lock_guard<std::mutex> lock(ctx->mutex);
        sdfgpu_set_option(ctx->handle, "host_pack", 0);
        const int rc = sdfgpu_build(ctx->handle, mask.data(), n, n, n, resolution, 0, ref.data(), &rmax, &rmin);
        sdfgpu_set_option(ctx->handle, "host_pack", 1);
        if (rc != SDFGPU_OK) { std::printf("FAIL: sdfgpu_build: %s\n", sdfgpu_last_error(ctx->handle)); return 1; }
    }
    const std::vector<float>& got = last.first.GetImmutableRawData();
    if (got.size() != N || std::memcmp(got.data(), ref.data(), N * sizeof(float)) != 0) { std::printf("FAIL: the class seam's field differs from the C ABI's\n"); return 1; }
    if (last.second.first != rmax || last.second.second != rmin) { std::printf("FAIL: extrema differ\n"); return 1; }
    double best = ms[0];
    for (double v : ms) best = v < best ? v : best;
    std::printf("{\"call\": \"CollisionMapGrid::ExtractSignedDistanceField(oob, true, false)\", \"n\": %lld, \"reps\": %d, \"min_ms\": %.3f, \"all_ms\": [",
                (long long)n, reps, best);
    for (size_t i = 0; i < ms.size(); ++i) std::printf("%s%.3f", i ? ", " : "", ms[i]);
    std::printf("], \"bit_identical_to_sdfgpu_build\": true, \"extrema\": [%.9g, %.9g], \"cells_MiB\": %.0f, \"field_MiB\": %.0f}\n", rmax, rmin,
                (double)N * 8 / 1048576.0, (double)N * 4 / 1048576.0);
    std::printf("class seam OK\n");
    return 0;
}
