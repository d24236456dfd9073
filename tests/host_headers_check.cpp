// Sanitizer harness of the CPU suite (SURVEY.md section 5 "ASan/UBSan for the oracle tests"; VERDICT r4 "next round" 7b).
// One binary, built twice by tests/test_host_cpu.py -- plain, and with -fsanitize=address,undefined -fno-sanitize-recover --
// that drives everything on the path that runs on the HOST and needs no GPU:
//   * the oracle's C restatement (oracle/sdf_oracle.c: reference algorithm, virtual border, exact EDT, brute force) on small
//     scenes, cross-checked against each other;
//   * the dense tier's policy object (sdf_tools_amd/csrc/sdfgpu_policy.hpp) through a build / report sequence;
//   * the mirror headers' host logic (include/sdf_tools, include/arc_utilities): the uninitialised result storage of the
//     build seams, moves that must not copy, serialisation round trips, hostile headers, per-point queries.
// TEST INFRASTRUCTURE.  Linked against libsdfgpu.so only because the inline seams of the headers name its symbols; no ABI call
// is reached (there is no GPU in the CPU suite).
// Prints "host headers OK" and returns 0; any sanitizer report aborts with a non-zero status.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

extern "C" {
int sdf_oracle_extract(const uint8_t* filled, int64_t nx, int64_t ny, int64_t nz, double resolution, float* out_sdf,
                       double* out_extrema, double* out_dsq_filled, double* out_dsq_free);
int sdf_oracle_extract_vb(const uint8_t* filled, int64_t nx, int64_t ny, int64_t nz, double resolution, int add_virtual_border,
                          float* out_sdf, double* out_extrema);
void sdf_oracle_classify_cells(const void* cells, int64_t n, int unknown_is_filled, uint8_t* out_mask);
int sdf_oracle_exact_edt(const uint8_t* seed, int seed_value, int64_t nx, int64_t ny, int64_t nz, int64_t* out);
int sdf_oracle_brute_edt(const uint8_t* seed, int seed_value, int64_t nx, int64_t ny, int64_t nz, int64_t* out);
int sdf_oracle_exact_sdf(const uint8_t* filled, int64_t nx, int64_t ny, int64_t nz, double resolution, int add_virtual_border,
                         float* out_sdf, double* out_extrema, int64_t* out_dsq);
}

#include "../sdf_tools_amd/csrc/sdfgpu_policy.hpp"
#include "sdf_tools/collision_map.hpp"
#include "sdf_tools/sdf.hpp"
#include "sdf_tools/tagged_object_collision_map.hpp"

#define CHECK(cond)                                                                          \
    do {                                                                                     \
        if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
    } while (0)

static int check_oracle() {
    std::mt19937 rng(11);
    const int64_t shapes[][3] = {{12, 9, 7}, {20, 40, 1}, {1, 1, 17}, {8, 8, 8}, {5, 1, 6}};
    for (const auto& s : shapes) {
        const int64_t nx = s[0], ny = s[1], nz = s[2], n = nx * ny * nz;
        for (double p : {0.5, 0.05, 0.0, 1.0}) {
            std::vector<uint8_t> m((size_t)n);
            for (auto& v : m) v = (rng() % 1000) < p * 1000 ? 1 : 0;
            std::vector<float> ref((size_t)n), ex((size_t)n), vb((size_t)n), exvb((size_t)n);
            double e1[2], e2[2], e3[2], e4[2];
            std::vector<int64_t> dsq((size_t)n), ed((size_t)n), bd((size_t)n);
            CHECK(sdf_oracle_extract(m.data(), nx, ny, nz, 0.25, ref.data(), e1, nullptr, nullptr) == 0);
            CHECK(sdf_oracle_exact_sdf(m.data(), nx, ny, nz, 0.25, 0, ex.data(), e2, dsq.data()) == 0);
            CHECK(sdf_oracle_extract_vb(m.data(), nx, ny, nz, 0.25, 1, vb.data(), e3) == 0);
            CHECK(sdf_oracle_exact_sdf(m.data(), nx, ny, nz, 0.25, 1, exvb.data(), e4, nullptr) == 0);
            for (int seed = 0; seed < 2; ++seed) {              // lower-envelope EDT against the O(N^2) brute force
                CHECK(sdf_oracle_exact_edt(m.data(), seed, nx, ny, nz, ed.data()) == 0);
                CHECK(sdf_oracle_brute_edt(m.data(), seed, nx, ny, nz, bd.data()) == 0);
                CHECK(ed == bd);
            }
            for (int64_t i = 0; i < n; ++i) {
                // the reference's propagation never under-estimates and is exact below d^2 = 8
                CHECK(std::fabs(ref[(size_t)i]) >= std::fabs(ex[(size_t)i]));
                const int64_t D = dsq[(size_t)i] < 0 ? -dsq[(size_t)i] : dsq[(size_t)i];
                if (D < 8) CHECK(ref[(size_t)i] == ex[(size_t)i]);
                CHECK(std::signbit(ref[(size_t)i]) == (m[(size_t)i] != 0) || ref[(size_t)i] == 0.0f || std::isinf(ref[(size_t)i]));
                CHECK(std::fabs(vb[(size_t)i]) >= std::fabs(exvb[(size_t)i]));
            }
        }
    }
    // the cell predicate (collision_map.hpp:689-704)
    struct Cell { float occ; uint32_t comp; };
    const Cell cells[6] = {{0.0f, 0}, {0.5f, 1}, {0.50001f, 2}, {1.0f, 3}, {NAN, 4}, {-1.0f, 5}};
    uint8_t m0[6], m1[6];
    sdf_oracle_classify_cells(cells, 6, 0, m0);
    sdf_oracle_classify_cells(cells, 6, 1, m1);
    const uint8_t w0[6] = {0, 0, 1, 1, 0, 0}, w1[6] = {0, 1, 1, 1, 0, 0};
    CHECK(std::memcmp(m0, w0, 6) == 0 && std::memcmp(m1, w1, 6) == 0);
    return 0;
}

static int check_policy() {
    sdfgpu::DensePolicy pol;
    for (int b = 0; b < 200; ++b) {
        const sdfgpu::DensePlan plan = pol.plan(true, false, true, false);
        pol.prev = sdfgpu::ReportedBuild{plan.dense, false, plan.fix_mode_build(), plan.staged};
        // a scene that certifies for a while, then stops, then certifies again
        const bool fails = b >= 40 && b < 120;
        if (plan.dense && (b % 3) != 1) pol.consume_report(fails, fails && plan.fix_mode_build(), fails);
    }
    const sdfgpu::DensePlan last = pol.plan(true, false, true, false);
    CHECK(last.dense || !last.dense);                         // (any state is legal; the sanitizers watch the arithmetic)
    // the far-field habit (round 5): a long run of far-field reports, re-probes, a scene change, every mode
    sdfgpu::FarHabit far;
    int predicted = 0;
    for (int b = 0; b < 5000; ++b) {
        const bool p = far.plan((b % 7) != 3, (b % 11) == 5);
        predicted += p ? 1 : 0;
        if (!p) far.consume_report(b < 3000 || (b % 2) == 0, b < 3000);
        if (b == 4000) far.set_mode(2);
        if (b == 4500) far.set_mode(0);
    }
    CHECK(predicted > 2000 && far.streak >= 0 && far.seq == 5000);
    return 0;
}

static int check_headers() {
    using sdf_tools::SignedDistanceField;
    // ---- the build seams' result storage: sized, writable everywhere, owned and freed by the vector --------------------------
    {
        std::vector<float> v = sdf_tools::detail::UninitializedFloatVector(100003);
        CHECK(v.size() == 100003 && v.capacity() >= v.size());
        for (size_t i = 0; i < v.size(); ++i) v[i] = (float)i;
        std::vector<float> w = v;                               // copy, then grow the original: ordinary vector behaviour
        v.push_back(-1.0f);
        CHECK(w.size() == 100003 && v.size() == 100004 && w[100002] == 100002.0f && v[100003] == -1.0f);
        CHECK(sdf_tools::detail::UninitializedFloatVector(0).empty());
    }
    SignedDistanceField a(SignedDistanceField::ForBuild{}, Eigen::Isometry3d::Identity(), "frame", 0.5, (int64_t)7, (int64_t)5, (int64_t)3, 99.0f);
    CHECK(a.GetImmutableRawData().size() == 105 && a.GetOOBValue() == 99.0f && a.GetNumZCells() == 3 && a.IsInitialized());
    float* d = a.MutableDataForBuild();
    for (int i = 0; i < 105; ++i) d[i] = 0.25f * (float)i - 3.0f;
    // ---- moves must move (the user-provided virtual destructor of VoxelGrid used to turn them into 512 MiB copies) -------------
    const float* before = a.GetImmutableRawData().data();
    SignedDistanceField b(std::move(a));
    CHECK(b.GetImmutableRawData().data() == before && a.GetImmutableRawData().empty());
    SignedDistanceField c;
    c = std::move(b);
    CHECK(c.GetImmutableRawData().data() == before && b.GetImmutableRawData().empty());
    auto pr = std::make_pair(std::move(c), std::make_pair(1.0, -1.0));
    CHECK(pr.first.GetImmutableRawData().data() == before);
    SignedDistanceField& s = pr.first;
    CHECK(s.GetImmutable((int64_t)6, (int64_t)4, (int64_t)2).first == d[104] && !s.GetImmutable((int64_t)7, (int64_t)0, (int64_t)0).second);
    // ---- locked fields refuse writes; per-point queries stay inside the array ----------------------------------------------
    s.Lock();
    CHECK(!s.SetValue((int64_t)1, (int64_t)1, (int64_t)1, 5.0f) && s.MutableDataForBuild() == nullptr);
    s.Unlock();
    CHECK(s.SetValue((int64_t)1, (int64_t)1, (int64_t)1, 5.0f));
    for (double x : {-1.0, 0.01, 1.7, 3.49, 3.51})
        for (double y : {-0.2, 0.3, 2.49})
            for (double z : {0.0, 0.7, 1.49, 2.0}) {
                (void)s.EstimateDistance(x, y, z);
                (void)s.GetGradient(x, y, z, true);
            }
    // ---- serialisation round trips and a hostile header ---------------------------------------------------------------------
    std::vector<uint8_t> buf;
    s.SerializeSelf(buf);
    SignedDistanceField back;
    CHECK(back.DeserializeSelf(buf, 0) == buf.size());
    CHECK(back.GetImmutableRawData() == s.GetImmutableRawData() && back.GetFrame() == "frame");
    std::vector<uint8_t> cut(buf.begin(), buf.begin() + (long)buf.size() / 2);
    bool threw = false;
    try { SignedDistanceField t; t.DeserializeSelf(cut, 0); } catch (const std::exception&) { threw = true; }
    CHECK(threw);
    sdf_tools::CollisionMapGrid map(Eigen::Isometry3d::Identity(), "world", 0.1, (int64_t)4, (int64_t)3, (int64_t)2, sdf_tools::COLLISION_CELL(0.0f));
    map.SetValue((int64_t)1, (int64_t)2, (int64_t)1, sdf_tools::COLLISION_CELL(1.0f, 7u));
    const sdf_tools::COLLISION_CELL* cells_before = map.GetImmutableRawData().data();
    sdf_tools::CollisionMapGrid moved(std::move(map));
    CHECK(moved.GetImmutableRawData().data() == cells_before && map.GetImmutableRawData().empty());
    const sdf_tools::CollisionMap msg = sdf_tools::CollisionMapGrid::GetMessageRepresentation(moved);
    const sdf_tools::CollisionMapGrid again = sdf_tools::CollisionMapGrid::LoadFromMessageRepresentation(msg);
    CHECK(again.GetImmutable((int64_t)1, (int64_t)2, (int64_t)1).first.component == 7u && again.GetFrame() == "world");
    sdf_tools::TaggedObjectCollisionMapGrid tagged(Eigen::Isometry3d::Identity(), "world", 0.1, (int64_t)3, (int64_t)3, (int64_t)3, sdf_tools::TAGGED_OBJECT_COLLISION_CELL());
    std::vector<uint8_t> tb;
    tagged.SerializeSelf(tb);
    sdf_tools::TaggedObjectCollisionMapGrid tback;
    CHECK(tback.DeserializeSelf(tb, 0) == tb.size() && tback.GetNumXCells() == 3);
    return 0;
}

int main() {
    if (check_oracle()) return 1;
    if (check_policy()) return 1;
    if (check_headers()) return 1;
    std::printf("host headers OK\n");
    return 0;
}
