// C++ drop-in check: the reference's tutorial scene (src/sdf_tools_tutorial.cpp:23-59,128-135) written
// against this repo's mirror headers -- same class names, constructors and method calls as the reference,
// linked against libsdfgpu.so.  Prints the values SURVEY.md section 4 pins for this scene and exits non-zero
// if they differ.  Build: see tests/test_cpp_example.py (g++ -std=c++17 -I include ... -lsdfgpu -lz).
#include <cmath>
#include <cstdio>
#include <stdexcept>

#include "sdf_tools/collision_map.hpp"
#include "sdf_tools/sdf.hpp"
#include "sdf_tools/tagged_object_collision_map.hpp"

int main(int argc, char** argv) {
    const bool compile_only_smoke = argc > 1 && std::string(argv[1]) == "--no-gpu";
    const double resolution = 0.25;
    Eigen::Isometry3d origin_transform = Eigen::Isometry3d::Identity();
#if SDF_TOOLS_HAVE_EIGEN
    origin_transform.translation() = Eigen::Vector3d(-5.0, -5.0, -5.0);
#else
    origin_transform.setTranslation(-5.0, -5.0, -5.0);
#endif
    const std::string frame = "tutorial_frame";
    sdf_tools::COLLISION_CELL oob_cell;
    oob_cell.occupancy = 0.0;
    oob_cell.component = 0;
    // 10 m cube at 0.25 m: the metric-size constructor gives 40 x 40 x 40 cells
    sdf_tools::CollisionMapGrid collision_map(origin_transform, frame, resolution, 10.0, 10.0, 10.0, oob_cell);
    if (collision_map.GetNumXCells() != 40) { std::printf("unexpected cell count\n"); return 2; }
    for (int64_t x = 0; x < collision_map.GetNumXCells(); x++)
        for (int64_t y = 0; y < collision_map.GetNumYCells(); y++)
            for (int64_t z = 0; z < collision_map.GetNumZCells(); z++)
                if (x < collision_map.GetNumXCells() / 2 && y < collision_map.GetNumYCells() / 2 && z < collision_map.GetNumZCells() / 2)
                    collision_map.SetValue(x, y, z, sdf_tools::COLLISION_CELL(1.0));
    // location-based access follows the grid convention (cell index = floor((p - origin) / resolution))
    const auto q = collision_map.GetImmutable(-4.9, -4.9, -4.9);
    if (!q.second || q.first.occupancy != 1.0f) { std::printf("location lookup failed\n"); return 2; }
    if (compile_only_smoke) {
        try {
            collision_map.ExtractSignedDistanceField(1e6f, true, false);
        } catch (const std::runtime_error& e) {
            std::printf("no GPU: %s\n", e.what());   // expected on a CPU-only box: the path has no CPU fallback
            return 0;
        }
    }
    const float oob_value = INFINITY;
    const auto sdf_with_extrema = collision_map.ExtractSignedDistanceField(oob_value, false, false);
    const sdf_tools::SignedDistanceField& sdf = sdf_with_extrema.first;
    const double max_d = sdf_with_extrema.second.first, min_d = sdf_with_extrema.second.second;
    const float a = sdf.GetImmutable((int64_t)10, (int64_t)10, (int64_t)10).first;
    const float b = sdf.GetImmutable((int64_t)20, (int64_t)20, (int64_t)20).first;
    const float c = sdf.GetImmutable((int64_t)39, (int64_t)39, (int64_t)39).first;
    std::printf("sdf(10,10,10)=%.7g sdf(20,20,20)=%.7g sdf(39,39,39)=%.7g extrema=(%.6g, %.6g)\n", a, b, c, max_d, min_d);
    const std::vector<double> g = sdf.GetGradient((int64_t)25, (int64_t)10, (int64_t)10, true);
    std::printf("gradient(25,10,10) = (%g, %g, %g)\n", g[0], g[1], g[2]);
    const auto est = sdf.EstimateDistance(2.6, -2.4, -2.4);
    std::printf("EstimateDistance = %g (%d)\n", est.first, (int)est.second);
    bool ok = a == -2.5f && std::fabs(b - 0.4330127f) < 1e-7 && std::fabs(c - 8.6602545f) < 1e-6 &&
              std::fabs(max_d - 8.660254037844387) < 1e-12 && min_d == -5.0 && g.size() == 3 && std::fabs(g[0] - 1.0) < 1e-6;
    // virtual border and the generic predicate seam give consistent results
    const auto vb = collision_map.ExtractSignedDistanceField(oob_value, false, true);
    const auto via_pred = collision_map.ExtractSignedDistanceFieldViaPredicate(oob_value, false, false);
    ok = ok && via_pred.first.GetImmutableRawData() == sdf.GetImmutableRawData() && vb.second.first <= max_d;
    // the cell-predicate overload (reference sdf_generation.hpp:422-441), as client code calls it
    const std::function<bool(const sdf_tools::COLLISION_CELL&)> cell_is_filled = [](const sdf_tools::COLLISION_CELL& cell) {
        return cell.occupancy > 0.5f;
    };
    const auto via_cell_pred = sdf_generation::ExtractSignedDistanceField<sdf_tools::COLLISION_CELL>(collision_map, cell_is_filled, oob_value, frame);
    ok = ok && via_cell_pred.first.GetImmutableRawData() == sdf.GetImmutableRawData() &&
         via_cell_pred.second.first == max_d && via_cell_pred.second.second == min_d;
    // full-grid gradient through the GPU fast path == the per-voxel host calls
    {
        const std::vector<double> flat = sdf.GetFullGradientFlat(true, 0.0);
        bool same = flat.size() == (size_t)40 * 40 * 40 * 3;
        for (int64_t x = 0; same && x < 40; x += 13)
            for (int64_t y = 0; y < 40; y += 7)
                for (int64_t z = 0; z < 40; z++) {
                    const std::vector<double> gh = sdf.GetGradient(x, y, z, true);
                    const size_t i = (size_t)((x * 40 + y) * 40 + z) * 3;
                    same = same && gh.size() == 3 && gh[0] == flat[i] && gh[1] == flat[i + 1] && gh[2] == flat[i + 2];
                }
        ok = ok && same;
    }
    // file round trip (SDFZ)
    sdf_tools::SignedDistanceField::SaveToFile(sdf, "/tmp/tutorial.sdf", true);
    const sdf_tools::SignedDistanceField back = sdf_tools::SignedDistanceField::LoadFromFile("/tmp/tutorial.sdf");
    ok = ok && back.GetImmutableRawData() == sdf.GetImmutableRawData() && back.GetFrame() == frame;
    // tagged-object map (reference tagged_object_collision_map.hpp): object 1 = the same box, object 2 = one far cell,
    // and one occupied cell without an object id
    const sdf_tools::TAGGED_OBJECT_COLLISION_CELL free_cell(0.0f, 0u);
    sdf_tools::TaggedObjectCollisionMapGrid tagged(origin_transform, frame, resolution, 40, 40, 40, free_cell);
    for (int64_t x = 0; x < 20; x++)
        for (int64_t y = 0; y < 20; y++)
            for (int64_t z = 0; z < 20; z++) tagged.SetValue(x, y, z, sdf_tools::TAGGED_OBJECT_COLLISION_CELL(1.0f, 1u));
    tagged.SetValue(30, 30, 30, sdf_tools::TAGGED_OBJECT_COLLISION_CELL(1.0f, 2u));
    tagged.SetValue(35, 5, 5, sdf_tools::TAGGED_OBJECT_COLLISION_CELL(1.0f, 0u));
    const auto only_box = tagged.ExtractSignedDistanceField(oob_value, std::vector<uint32_t>{1u}, false, false);
    ok = ok && only_box.first.GetImmutableRawData() == sdf.GetImmutableRawData();      // object filter: the box alone
    const auto all_objects = tagged.ExtractSignedDistanceField(oob_value, std::vector<uint32_t>{}, false, false);
    const auto combined = tagged.ExtractFreeAndNamedObjectsSignedDistanceField(oob_value, false);
    ok = ok && combined.first.GetImmutable((int64_t)35, (int64_t)5, (int64_t)5).first == 0.0f            // unnamed filled
            && combined.first.GetImmutable((int64_t)30, (int64_t)30, (int64_t)30).first == -0.25f        // named, 1 cell
            && combined.first.GetImmutable((int64_t)34, (int64_t)5, (int64_t)5).first == 0.25f           // free, next to it
            && combined.first.GetImmutable((int64_t)10, (int64_t)10, (int64_t)10).first == -2.5f
            && all_objects.first.GetImmutable((int64_t)35, (int64_t)5, (int64_t)5).first == -0.25f;
    const auto per_object = tagged.MakeAllObjectSDFs(false, false);
    ok = ok && per_object.size() == 2 && per_object.at(1u).GetImmutableRawData() == sdf.GetImmutableRawData() &&
         per_object.at(2u).GetImmutable((int64_t)30, (int64_t)30, (int64_t)30).first == -0.25f &&
         per_object.at(2u).GetImmutable((int64_t)10, (int64_t)10, (int64_t)10).first > 0.0f;
    std::printf(ok ? "tutorial scene OK\n" : "tutorial scene MISMATCH\n");
    return ok ? 0 : 1;
}
