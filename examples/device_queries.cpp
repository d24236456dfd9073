// C++ client of the device-resident seam (round 4, next-row N1 for C++ callers): build a CollisionMapGrid scene, leave
// the signed distance field in HBM (CollisionMapGrid::ExtractSignedDistanceFieldDevice), answer n EstimateDistance +
// GetGradient queries with one kernel -- without the nx*ny*nz*4-byte download the host seam pays before the first query
// -- then download lazily and check a sample of the answers against the reference-shaped host calls
// (SignedDistanceField::EstimateDistance3d / GetGradient3d, reference include/sdf_tools/sdf.hpp:947-953, :395-403).
//
//   device_queries [n_cells = 128] [n_points = 1048576] [--no-gpu]
// Build: see tests/test_cpp_example.py (g++ -std=c++17 -I include ... -lsdfgpu -lz).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "sdf_tools/collision_map.hpp"
#include "sdf_tools/device_sdf.hpp"
#include "sdf_tools/sdf.hpp"

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
    int64_t n = 128, n_points = 1 << 20;
    bool no_gpu = false;
    int pos = 0;
    for (int i = 1; i < argc; ++i) {
        if (std::string(argv[i]) == "--no-gpu") no_gpu = true;
        else if (pos++ == 0) n = std::atoll(argv[i]);
        else n_points = std::atoll(argv[i]);
    }
    const double resolution = 0.01;
    Eigen::Isometry3d origin = Eigen::Isometry3d::Identity();
#if SDF_TOOLS_HAVE_EIGEN
    origin.translation() = Eigen::Vector3d(-0.5, 0.25, 0.0);
#else
    origin.setTranslation(-0.5, 0.25, 0.0);
#endif
    sdf_tools::CollisionMapGrid map(origin, "world", resolution, n, n, n, sdf_tools::COLLISION_CELL(0.0));
    // two boxes in free space (the pattern of the reference's scripts/3d_sdf_demo_rviz.py:15-19, scaled to the grid)
    for (int64_t x = n / 2; x < n / 2 + n / 5; ++x)
        for (int64_t y = n / 2; y < n / 2 + n / 10; ++y)
            for (int64_t z = 0; z < n / 2; ++z) map.SetValue(x, y, z, sdf_tools::COLLISION_CELL(1.0));
    for (int64_t x = n / 2; x < n / 2 + n / 4; ++x)
        for (int64_t y = n / 5; y < 2 * n / 5; ++y)
            for (int64_t z = n / 4; z < n / 2; ++z) map.SetValue(x, y, z, sdf_tools::COLLISION_CELL(1.0));
    if (no_gpu) {
        try {
            map.ExtractSignedDistanceFieldDevice(INFINITY, false, false);
        } catch (const std::runtime_error& e) {
            std::printf("no GPU: %s\n", e.what());               // expected on a CPU-only box: no CPU fallback
            return 0;
        }
        std::printf("a GPU is present\n");
        return 0;
    }
    // world-frame query points: inside the grid, a few outside
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> u(-0.02, n * resolution + 0.02);
    std::vector<double> pts((size_t)n_points * 3);
    for (int64_t i = 0; i < n_points; ++i) {
        pts[3 * i] = -0.5 + u(rng);
        pts[3 * i + 1] = 0.25 + u(rng);
        pts[3 * i + 2] = 0.0 + u(rng);
    }
    std::vector<double> dist((size_t)n_points), grad((size_t)n_points * 3);
    std::vector<uint8_t> flags((size_t)n_points);

    map.ExtractSignedDistanceFieldDevice(INFINITY, false, false);     // warm-up: context creation, allocations
    const double t0 = now_ms();
    auto built = map.ExtractSignedDistanceFieldDevice(INFINITY, false, false);
    const double t1 = now_ms();
    sdf_tools::DeviceSignedDistanceField& dsdf = built.first;
    dsdf.QueryBatch(pts.data(), n_points, true, dist.data(), grad.data(), flags.data());
    const double t2 = now_ms();
    if (dsdf.HostCopyExists()) { std::printf("FAIL: the field was downloaded before anybody asked for it\n"); return 1; }
    std::printf("%lld^3 cells: build to device %.3f ms, %lld EstimateDistance + GetGradient queries %.3f ms, no field download (%.1f MiB stayed in HBM)\n",
                (long long)n, t1 - t0, (long long)n_points, t2 - t1, (double)(n * n * n) * 4.0 / 1048576.0);

    // the host seam for comparison: the same build with the download, then the reference-shaped per-point calls
    const double t3 = now_ms();
    const auto host_built = map.ExtractSignedDistanceField(INFINITY, false, false);
    const double t4 = now_ms();
    std::printf("host seam: ExtractSignedDistanceField (with download) %.3f ms\n", t4 - t3);
    const sdf_tools::SignedDistanceField& hs = host_built.first;
    bool ok = built.second == host_built.second && dsdf.GetExtrema() == host_built.second;
    // lazy download == the host seam's field
    ok = ok && dsdf.Host().GetImmutableRawData() == hs.GetImmutableRawData() && dsdf.HostCopyExists();
    int64_t checked = 0, inside = 0;
    double worst = 0.0;
    const int64_t stride = n_points > 20000 ? n_points / 20000 : 1;
    for (int64_t i = 0; i < n_points; i += stride) {
        const Eigen::Vector3d p(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
        const std::pair<double, bool> e = hs.EstimateDistance3d(p);
        const std::vector<double> g = hs.GetGradient3d(p, true);
        ok = ok && e.second == ((flags[i] & 1u) != 0) && (g.size() == 3) == ((flags[i] & 2u) != 0);
        if (e.second) {
            ++inside;
            if (std::isfinite(e.first)) worst = std::fmax(worst, std::fabs(e.first - dist[i]));
            else ok = ok && e.first == dist[i];
        } else ok = ok && std::isinf(dist[i]);                     // OOB value
        if (g.size() == 3)
            for (int k = 0; k < 3; ++k) {
                if (std::isfinite(g[k])) worst = std::fmax(worst, std::fabs(g[k] - grad[3 * i + k]));
                else ok = ok && (std::isnan(g[k]) ? std::isnan(grad[3 * i + k]) : g[k] == grad[3 * i + k]);
            }
        ++checked;
    }
    // the reference-shaped vector forms agree with the flat one
    {
        std::vector<Eigen::Vector3d> some;
        for (int64_t i = 0; i < 64 && i < n_points; ++i) some.emplace_back(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
        const auto ed = dsdf.EstimateDistanceBatch(some);
        const auto gd = dsdf.GetGradientBatch(some, true);
        for (size_t i = 0; i < some.size(); ++i) {
            ok = ok && ed[i].second == ((flags[i] & 1u) != 0) && (ed[i].first == dist[i] || (std::isnan(ed[i].first) && std::isnan(dist[i])));
            ok = ok && (gd[i].size() == 3) == ((flags[i] & 2u) != 0);
        }
    }
    // a host field uploaded again answers the same
    {
        sdf_tools::DeviceSignedDistanceField up = sdf_tools::DeviceSignedDistanceField::Upload(hs);
        std::vector<double> d2(1024);
        const int64_t m = n_points < 1024 ? n_points : 1024;
        up.QueryBatch(pts.data(), m, true, d2.data(), nullptr, nullptr);
        for (int64_t i = 0; i < m; ++i) ok = ok && (d2[i] == dist[i] || (std::isnan(d2[i]) && std::isnan(dist[i])));
    }
    std::printf("checked %lld points against the host calls (%lld inside the grid): max |difference| = %.3g\n",
                (long long)checked, (long long)inside, worst);
    ok = ok && worst <= 1e-9 && inside > checked / 2;
    std::printf(ok ? "device queries OK\n" : "device queries MISMATCH\n");
    return ok ? 0 : 1;
}
