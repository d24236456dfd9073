// SignedDistanceField -- result container of the SDF build path, API-compatible with the subset
// of sdf_tools::SignedDistanceField (reference include/sdf_tools/sdf.hpp) that the hot path and the
// pysdf_tools surface use: constructors (:34-81), lock-guarded SetValue (:236-264), GetGradient /
// GetGridAlignedGradient / GetFullGradient (:341-526), EstimateDistance (:699-961), serialisation
// and file / message forms (src/sdf_tools/sdf.cpp:213-502).  Out of scope here:
// local-extrema maps, projection out of collision, AutoDiff gradients, RViz export.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "arc_utilities/serialization.hpp"
#include "arc_utilities/voxel_grid.hpp"
#include "arc_utilities/zlib_helpers.hpp"
#include "sdf_tools/eigen_lite.hpp"
#include "sdf_tools/gpu_context.hpp"

namespace sdf_tools {
using VoxelGrid::GRID_INDEX;

// Plain mirror of msg/SDF.msg for builds without ROS (field names kept).
struct SDF {
    struct Header { uint32_t seq = 0; double stamp = 0.0; std::string frame_id; } header;
    std::vector<uint8_t> serialized_sdf;
    bool is_compressed = false;
};

namespace detail {
// A std::vector<float> of n elements whose storage nobody has written: the pages of a 512^3 field (512 MiB) are then first
// touched by the thread team that drains the device-to-host copy into it -- in parallel, once -- instead of by one thread
// value-initialising 134 M floats that the drain overwrites (measured in round 4: the largest part of a 241 ms class call
// around 0.14 ms of kernels).  The container type must stay std::vector<float> (GetImmutableRawData() returns it by
// reference, reference include/sdf_tools/sdf.hpp / VoxelGrid), and the standard offers no way to size one without writing
// its elements, so on libstdc++ the three pointers of an empty vector are pointed at memory from its own allocator
// (begin, end, end of storage -- a layout that has not changed since GCC 3; checked against a normally built vector at run
// time); anywhere else, or if the check fails, the vector is built the ordinary way and only speed is lost.
inline std::vector<float> UninitializedFloatVector(const size_t n) {
    // OPT-IN (round 6): the trick overwrites a std::vector's representation -- undefined behaviour by the letter of the standard, and
    // invisible to an instrumented vector -- so it is compiled only when the client defines SDF_TOOLS_VECTOR_ADOPT (the in-tree
    // builds of pysdf_tools and the examples do; INTEGRATION.md's drop-in recipe does not), never in debug / sanitizer builds of
    // libstdc++.  Without it the storage is value-initialised the ordinary way: +75 ... 100 ms per 512^3 result (measured: profiles/r06_redzone_suite_and_fuzz.txt).
#if defined(SDF_TOOLS_VECTOR_ADOPT) && defined(__GLIBCXX__) && !defined(_GLIBCXX_DEBUG) && !defined(_GLIBCXX_SANITIZE_VECTOR) && \
    !defined(__SANITIZE_ADDRESS__) && !defined(__SANITIZE_THREAD__) && !defined(SDF_TOOLS_NO_VECTOR_ADOPT)
    static_assert(sizeof(std::vector<float>) == 3 * sizeof(float*), "std::vector<float> is not three pointers");
    static const bool layout_ok = []() {
        std::vector<float> probe(3);
        probe.reserve(5);
        float* rep[3];
        std::memcpy(rep, static_cast<const void*>(&probe), sizeof rep);
        return rep[0] == probe.data() && rep[1] == probe.data() + probe.size() && rep[2] == probe.data() + probe.capacity();
    }();
    if (layout_ok && n > 0) {
        std::vector<float> v;
        float* const p = std::allocator<float>().allocate(n);      // (operator new: untouched pages for large n)
        float* rep[3] = {p, p + n, p + n};
        std::memcpy(static_cast<void*>(&v), rep, sizeof rep);      // v now owns p; its destructor deallocates through the same allocator
        return v;
    }
#endif
    return std::vector<float>(n);
}
}  // namespace detail

class SignedDistanceField : public VoxelGrid::VoxelGrid<float> {
protected:
    std::string frame_;
    bool locked_;

public:
    // Tag of the constructor the build seams use: every cell is about to be overwritten by the device-to-host drain, so the
    // array is NOT filled with OOB_value first (detail::UninitializedFloatVector).  Cells are indeterminate until then.
    struct ForBuild {};
    SignedDistanceField(ForBuild, const Eigen::Isometry3d& origin_transform, const std::string& frame, double resolution,
                        int64_t x_cells, int64_t y_cells, int64_t z_cells, float OOB_value)
        : Base(), frame_(frame), locked_(false) {
        if (x_cells <= 0 || y_cells <= 0 || z_cells <= 0) throw std::invalid_argument("cell counts must be positive");
        std::vector<float> storage = detail::UninitializedFloatVector((size_t)(x_cells * y_cells * z_cells));
        Setup(origin_transform, resolution, resolution, resolution, x_cells, y_cells, z_cells, OOB_value, OOB_value, &storage);
    }

    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    typedef std::shared_ptr<SignedDistanceField> Ptr;
    typedef std::shared_ptr<const SignedDistanceField> ConstPtr;
    using Base = ::VoxelGrid::VoxelGrid<float>;

    SignedDistanceField(const std::string& frame, double resolution, double x_size, double y_size, double z_size, float OOB_value)
        : Base(resolution, x_size, y_size, z_size, OOB_value), frame_(frame), locked_(false) {}
    SignedDistanceField(const Eigen::Isometry3d& origin_transform, const std::string& frame, double resolution,
                        double x_size, double y_size, double z_size, float OOB_value)
        : Base(origin_transform, resolution, x_size, y_size, z_size, OOB_value), frame_(frame), locked_(false) {}
    SignedDistanceField(const std::string& frame, double resolution, int64_t x_cells, int64_t y_cells, int64_t z_cells, float OOB_value)
        : Base(resolution, x_cells, y_cells, z_cells, OOB_value), frame_(frame), locked_(false) {}
    SignedDistanceField(const Eigen::Isometry3d& origin_transform, const std::string& frame, double resolution,
                        int64_t x_cells, int64_t y_cells, int64_t z_cells, float OOB_value)
        : Base(origin_transform, resolution, x_cells, y_cells, z_cells, OOB_value), frame_(frame), locked_(false) {}
    SignedDistanceField() : Base(), frame_(""), locked_(false) {}

    Base* Clone() const override { return new SignedDistanceField(*this); }

    double GetResolution() const { return GetCellSizes().x(); }
    std::string GetFrame() const { return frame_; }
    void SetFrame(const std::string& f) { frame_ = f; }
    bool IsLocked() const { return locked_; }
    void Lock() { locked_ = true; }
    void Unlock() { locked_ = false; }

    // Writes are refused while locked (reference sdf.hpp:236-264).
    bool SetValue(const int64_t x, const int64_t y, const int64_t z, const float& value) override {
        if (IndexInBounds(x, y, z) && !locked_) { AccessIndex(GetDataIndex(x, y, z)) = value; return true; }
        return false;
    }
    bool SetValue(const GRID_INDEX& i, const float& value) override { return SetValue(i.x, i.y, i.z, value); }
    bool SetValue4d(const Eigen::Vector4d& l, const float& value) override { return SetValue(LocationToGridIndex4d(l), value); }
    bool SetValue3d(const Eigen::Vector3d& l, const float& value) override { return SetValue(LocationToGridIndex3d(l), value); }
    bool SetValue(const double x, const double y, const double z, const float& value) override {
        return SetValue(LocationToGridIndex(x, y, z), value);
    }
    // The build path writes the whole field at once (device -> host copy lands here).
    float* MutableDataForBuild() { return locked_ ? nullptr : data_.data(); }

    // ---- gradients (reference sdf.hpp:341-526) ----------------------------------------------
    std::vector<double> GetGridAlignedGradient(const int64_t x, const int64_t y, const int64_t z,
                                               const bool enable_edge_gradients = false) const {
        if (!IndexInBounds(x, y, z)) return std::vector<double>();
        const int64_t nx = GetNumXCells(), ny = GetNumYCells(), nz = GetNumZCells();
        auto at = [&](int64_t a, int64_t b, int64_t c) -> float { return AccessIndex(GetDataIndex(a, b, c)); };
        if (x > 0 && y > 0 && z > 0 && x < nx - 1 && y < ny - 1 && z < nz - 1) {
            // central differences: float subtraction, double scale (:447-458)
            const double inv_twice_resolution = 1.0 / (2.0 * GetResolution());
            const double gx = (at(x + 1, y, z) - at(x - 1, y, z)) * inv_twice_resolution;
            const double gy = (at(x, y + 1, z) - at(x, y - 1, z)) * inv_twice_resolution;
            const double gz = (at(x, y, z + 1) - at(x, y, z - 1)) * inv_twice_resolution;
            return std::vector<double>{gx, gy, gz};
        }
        if (!enable_edge_gradients) return std::vector<double>();
        // clamped one-sided differences on the boundary shell (:464-512)
        const int64_t lx = std::max<int64_t>(0, x - 1), hx = std::min(nx - 1, x + 1);
        const int64_t ly = std::max<int64_t>(0, y - 1), hy = std::min(ny - 1, y + 1);
        const int64_t lz = std::max<int64_t>(0, z - 1), hz = std::min(nz - 1, z + 1);
        const double ix = (double)(hx - lx) * GetResolution(), iy = (double)(hy - ly) * GetResolution(),
                     iz = (double)(hz - lz) * GetResolution();
        double gx = 0.0, gy = 0.0, gz = 0.0;
        if (ix > 0.0) gx = ((double)at(hx, y, z) - (double)at(lx, y, z)) * (1.0 / ix);
        if (iy > 0.0) gy = ((double)at(x, hy, z) - (double)at(x, ly, z)) * (1.0 / iy);
        if (iz > 0.0) gz = ((double)at(x, y, hz) - (double)at(x, y, lz)) * (1.0 / iz);
        return std::vector<double>{gx, gy, gz};
    }

    // Grid-aligned gradient rotated into the world frame (:405-430).
    std::vector<double> GetGradient(const int64_t x, const int64_t y, const int64_t z, const bool enable_edge_gradients = false) const {
        const std::vector<double> g = GetGridAlignedGradient(x, y, z, enable_edge_gradients);
        if (g.size() != 3) return std::vector<double>();
        const Eigen::Quaterniond q(origin_transform_.rotation());
        const Eigen::Quaterniond r = q * (Eigen::Quaterniond(0.0, g[0], g[1], g[2]) * q.inverse());
        return std::vector<double>{r.x(), r.y(), r.z()};
    }
    std::vector<double> GetGradient(const GRID_INDEX& i, const bool e = false) const { return GetGradient(i.x, i.y, i.z, e); }
    std::vector<double> GetGradient4d(const Eigen::Vector4d& l, const bool e = false) const {
        const GRID_INDEX i = LocationToGridIndex4d(l);
        return IndexInBounds(i) ? GetGradient(i, e) : std::vector<double>();
    }
    std::vector<double> GetGradient3d(const Eigen::Vector3d& l, const bool e = false) const {
        const GRID_INDEX i = LocationToGridIndex3d(l);
        return IndexInBounds(i) ? GetGradient(i, e) : std::vector<double>();
    }
    std::vector<double> GetGradient(const double x, const double y, const double z, const bool e = false) const {
        return GetGradient4d(Eigen::Vector4d(x, y, z, 1.0), e);
    }

    // Every cell's world-frame gradient at once, [x][y][z][3] doubles, computed by ONE kernel on the MI355X
    // (sdfgpu_gradient) instead of nx*ny*nz GetGradient calls: the values are those of GetGradient(x, y, z,
    // enable_edge_gradients) bit for bit (same float-subtract / double-scale arithmetic, :447-512); cells for which
    // the reference returns an empty vector hold `empty_fill` in all three components.  A rotated origin frame
    // applies the reference's quaternion sandwich (:405-430) on the host to the grid-aligned values.
    std::vector<double> GetFullGradientFlat(const bool enable_edge_gradients, const double empty_fill) const {
        const int64_t nx = GetNumXCells(), ny = GetNumYCells(), nz = GetNumZCells();
        std::vector<double> g((size_t)(nx * ny * nz) * 3);
        if (g.empty()) return g;
        const std::shared_ptr<sdf_generation::SharedGpuContext> ctx = sdf_generation::GpuContext::Shared();
        const std::lock_guard<std::mutex> lock(ctx->mutex);
        sdfgpu_handle h = ctx->handle;
        sdf_generation::ThrowOnStatus(h, sdfgpu_gradient(h, data_.data(), nx, ny, nz, GetResolution(),
                                                         enable_edge_gradients ? 1 : 0, g.data(), 1));
        const Eigen::Quaterniond q(origin_transform_.rotation());
        const bool identity = q.w() == 1.0 && q.x() == 0.0 && q.y() == 0.0 && q.z() == 0.0;
        const Eigen::Quaterniond qi = q.inverse();
        // The reference returns an EMPTY vector for a cell on the boundary shell when edge gradients are off (:464): those
        // cells get `empty_fill`.  Emptiness is decided from the index, not from the value: a genuine NaN (inf - inf on a grid
        // without filled or without free voxels) is what the reference returns there too and must stay.
        const int64_t sx = ny * nz;
        if (!identity) {
            for (size_t i = 0; i < g.size(); i += 3) {
                const Eigen::Quaterniond r = q * (Eigen::Quaterniond(0.0, g[i], g[i + 1], g[i + 2]) * qi);
                g[i] = r.x(); g[i + 1] = r.y(); g[i + 2] = r.z();
            }
        }                                                     // (identity frame: q * (0, g) * q^-1 == g exactly, nothing to do)
        if (!enable_edge_gradients) {
            // ... written over the six faces only (a loop over all cells with two divisions each cost far more than
            // the kernel -- 134 M iterations at 512^3 for 1.6 M shell cells)
            auto fill = [&](int64_t x, int64_t y, int64_t z) { double* c = &g[(size_t)(x * sx + y * nz + z) * 3]; c[0] = c[1] = c[2] = empty_fill; };
            for (int64_t x = 0; x < nx; ++x) {
                const bool xface = x == 0 || x == nx - 1;
                for (int64_t y = 0; y < ny; ++y) {
                    if (xface || y == 0 || y == ny - 1) { for (int64_t z = 0; z < nz; ++z) fill(x, y, z); }
                    else { fill(x, y, 0); fill(x, y, nz - 1); }
                }
            }
        }
        return g;
    }

    using GradientFunction = std::function<std::vector<double>(int64_t, int64_t, int64_t, bool)>;
    // Every cell's gradient via a caller-supplied function (:341-358); default cell = 3 x OOB value.
    ::VoxelGrid::VoxelGrid<std::vector<double>> GetFullGradient(const GradientFunction& gradient_function,
                                                              const bool enable_edge_gradients = false) const {
        ::VoxelGrid::VoxelGrid<std::vector<double>> grid(origin_transform_, GetResolution(), GetNumXCells(), GetNumYCells(),
                                                       GetNumZCells(), std::vector<double>(3, oob_value_));
        for (int64_t x = 0; x < GetNumXCells(); ++x)
            for (int64_t y = 0; y < GetNumYCells(); ++y)
                for (int64_t z = 0; z < GetNumZCells(); ++z)
                    grid.SetValue(x, y, z, gradient_function(x, y, z, enable_edge_gradients));
        return grid;
    }

    // ---- trilinear distance estimate (:699-961) ----------------------------------------------
protected:
    double CorrectedCenterDistance(const int64_t x, const int64_t y, const int64_t z) const {
        if (!IndexInBounds(x, y, z)) throw std::invalid_argument("Index out of bounds");
        const double d = (double)AccessIndex(GetDataIndex(x, y, z));
        const double half = GetResolution() * 0.5;              // shrink by half a cell toward the surface (:773-796)
        return d >= 0.0 ? d - half : d + half;
    }
    static std::pair<int64_t, int64_t> AxisInterpolationIndices(const int64_t i, const int64_t n, const double offset) {
        int64_t lower = i, upper = i;                            // (:798-833)
        if (offset >= 0.0) {
            upper = i + 1;
            if (upper >= n) { upper = i; lower = i - 1; if (lower < 0) lower = i; }
        } else {
            lower = i - 1;
            if (lower < 0) { upper = i + 1; lower = i; if (upper >= n) upper = i; }
        }
        return std::make_pair(lower, upper);
    }
    static double Bilinear(double l1, double h1, double l2, double h2, double q1, double q2, double ll, double lh, double hl, double hh) {
        // (multiplier * [h1-q1, q1-l1]) * [[ll, lh], [hl, hh]] * [h2-q2, q2-l2]^T, evaluated left to right (:699-727)
        const double multiplier = 1.0 / ((h1 - l1) * (h2 - l2));
        const double a0 = multiplier * (h1 - q1), a1 = multiplier * (q1 - l1);
        const double r0 = a0 * ll + a1 * hl, r1 = a0 * lh + a1 * hh;
        return r0 * (h2 - q2) + r1 * (q2 - l2);
    }
    double EstimateFromNeighborsGridFrame(const Eigen::Vector4d& q, const int64_t x, const int64_t y, const int64_t z) const {
        const Eigen::Vector4d c = GridIndexToLocationGridFrame(x, y, z);
        const auto xi = AxisInterpolationIndices(x, GetNumXCells(), q(0) - c(0));
        const auto yi = AxisInterpolationIndices(y, GetNumYCells(), q(1) - c(1));
        const auto zi = AxisInterpolationIndices(z, GetNumZCells(), q(2) - c(2));
        const Eigen::Vector4d lo = GridIndexToLocationGridFrame(xi.first, yi.first, zi.first);
        const double res = GetResolution();
        auto D = [&](int64_t a, int64_t b, int64_t cc) { return CorrectedCenterDistance(a, b, cc); };
        const double mz = Bilinear(lo(0), lo(0) + res, lo(1), lo(1) + res, q(0), q(1), D(xi.first, yi.first, zi.first),
                                   D(xi.first, yi.second, zi.first), D(xi.second, yi.first, zi.first), D(xi.second, yi.second, zi.first));
        const double pz = Bilinear(lo(0), lo(0) + res, lo(1), lo(1) + res, q(0), q(1), D(xi.first, yi.first, zi.second),
                                   D(xi.first, yi.second, zi.second), D(xi.second, yi.first, zi.second), D(xi.second, yi.second, zi.second));
        const double slope = (pz - mz) * (1.0 / res);            // (:745-771)
        return mz + ((q(2) - lo(2)) * slope);
    }

public:
    std::pair<double, bool> EstimateDistance4d(const Eigen::Vector4d& location) const {
        const GRID_INDEX i = LocationToGridIndex4d(location);
        if (!IndexInBounds(i)) return std::make_pair((double)GetOOBValue(), false);
        return std::make_pair(EstimateFromNeighborsGridFrame(inverse_origin_transform_ * location, i.x, i.y, i.z), true);
    }
    std::pair<double, bool> EstimateDistance3d(const Eigen::Vector3d& l) const { return EstimateDistance4d(Eigen::Vector4d(l.x(), l.y(), l.z(), 1.0)); }
    std::pair<double, bool> EstimateDistance(const double x, const double y, const double z) const {
        return EstimateDistance4d(Eigen::Vector4d(x, y, z, 1.0));
    }

    // ---- serialisation (src/sdf_tools/sdf.cpp:213-502) ------------------------------------------
    using FloatSerializer = std::function<uint64_t(const float&, std::vector<uint8_t>&)>;
    using FloatDeserializer = std::function<std::pair<float, uint64_t>(const std::vector<uint8_t>&, const uint64_t)>;

    uint64_t SerializeSelf(std::vector<uint8_t>& buffer,
                           const FloatSerializer& value_serializer = arc_utilities::SerializeFixedSizePOD<float>) const override {
        (void)value_serializer;
        const uint64_t start = buffer.size();
        BaseSerializeSelf(buffer, arc_utilities::SerializeFixedSizePOD<float>);
        arc_utilities::SerializeString(frame_, buffer);
        arc_utilities::SerializeFixedSizePOD<uint8_t>((uint8_t)locked_, buffer);
        return buffer.size() - start;
    }
    uint64_t DeserializeSelf(const std::vector<uint8_t>& buffer, const uint64_t current,
                             const FloatDeserializer& value_deserializer = arc_utilities::DeserializeFixedSizePOD<float>) override {
        (void)value_deserializer;
        uint64_t pos = current;
        pos += BaseDeserializeSelf(buffer, pos, arc_utilities::DeserializeFixedSizePOD<float>);
        const auto f = arc_utilities::DeserializeString(buffer, pos); pos += f.second;
        const auto l = arc_utilities::DeserializeFixedSizePOD<uint8_t>(buffer, pos); pos += l.second;
        frame_ = f.first;
        locked_ = (bool)l.first;
        return pos - current;
    }

    static void SaveToFile(const SignedDistanceField& sdf, const std::string& filepath, const bool compress) {
        std::vector<uint8_t> buffer;
        sdf.SerializeSelf(buffer);
        std::ofstream out(filepath, std::ios::out | std::ios::binary);
        const std::vector<uint8_t> body = compress ? ZlibHelpers::CompressBytes(buffer) : buffer;
        out.write(compress ? "SDFZ" : "SDFR", 4);                 // 4-byte magic (:392-416)
        out.write(reinterpret_cast<const char*>(body.data()), (std::streamsize)body.size());
    }
    static SignedDistanceField LoadFromFile(const std::string& filepath) {
        std::ifstream in(filepath, std::ios::in | std::ios::binary);
        if (!in.good()) throw std::invalid_argument("File does not exist");
        std::vector<uint8_t> all((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        if (all.size() < 4) throw std::invalid_argument("File is too small");
        const std::string magic(all.begin(), all.begin() + 4);
        std::vector<uint8_t> body(all.begin() + 4, all.end());
        SignedDistanceField sdf;
        if (magic == "SDFZ") sdf.DeserializeSelf(ZlibHelpers::DecompressBytes(body), 0);
        else if (magic == "SDFR") sdf.DeserializeSelf(body, 0);
        else throw std::invalid_argument("File has invalid header [" + magic + "]");
        return sdf;
    }
    static SDF GetMessageRepresentation(const SignedDistanceField& sdf) {
        SDF msg;                                                  // always zlib-compressed (:472-483)
        msg.header.frame_id = sdf.GetFrame();
        std::vector<uint8_t> buffer;
        sdf.SerializeSelf(buffer);
        msg.serialized_sdf = ZlibHelpers::CompressBytes(buffer);
        msg.is_compressed = true;
        return msg;
    }
    static SignedDistanceField LoadFromMessageRepresentation(const SDF& message) {
        SignedDistanceField sdf;
        if (message.is_compressed) sdf.DeserializeSelf(ZlibHelpers::DecompressBytes(message.serialized_sdf), 0);
        else sdf.DeserializeSelf(message.serialized_sdf, 0);
        return sdf;
    }
};

}  // namespace sdf_tools
