// Dense voxel grid container with the interface subset of UM-ARM-Lab/arc_utilities'
// VoxelGrid::VoxelGrid<T> that sdf_tools' kept classes inherit (SURVEY.md 8(b) "Inherited API").
// arc_utilities is not vendored by the reference, so this is an independent implementation of the
// conventions its callers rely on:
//   data index        = x * (ny*nz) + y * nz + z            (src/sdf_tools/utils_3d.py:71-73,
//                                                             strides serialised at sdf.cpp:241-245)
//   grid-frame cell i = floor(coordinate * inv_cell_size)
//   cell centre       = (i + 0.5) * cell_size
//   size constructor  : cells = ceil(size / resolution)
// The last three follow upstream arc_utilities and cannot be verified in this image (no copy of
// it exists here); they affect location<->index helpers only, never the distance computation.
#pragma once
#include <cmath>
#include <cstdint>
#include <functional>
#include <stdexcept>
#include <utility>
#include <vector>

#include "arc_utilities/serialization.hpp"
#include "sdf_tools/eigen_lite.hpp"

namespace VoxelGrid {

struct GRID_INDEX {
    int64_t x;
    int64_t y;
    int64_t z;
    GRID_INDEX() : x(-1), y(-1), z(-1) {}
    GRID_INDEX(const int64_t in_x, const int64_t in_y, const int64_t in_z) : x(in_x), y(in_y), z(in_z) {}
    bool operator==(const GRID_INDEX& o) const { return x == o.x && y == o.y && z == o.z; }
};

template <typename T, typename BackingStore = std::vector<T>>
class VoxelGrid {
protected:
    Eigen::Isometry3d origin_transform_;
    Eigen::Isometry3d inverse_origin_transform_;
    T default_value_;
    T oob_value_;
    BackingStore data_;
    double cell_x_size_ = 0.0, cell_y_size_ = 0.0, cell_z_size_ = 0.0;
    double inv_cell_x_size_ = 0.0, inv_cell_y_size_ = 0.0, inv_cell_z_size_ = 0.0;
    double x_size_ = 0.0, y_size_ = 0.0, z_size_ = 0.0;
    int64_t stride1_ = 0, stride2_ = 0;
    int64_t num_x_cells_ = 0, num_y_cells_ = 0, num_z_cells_ = 0;
    bool initialized_ = false;

    static int64_t CellsForSize(const double size, const double cell) { return (int64_t)std::ceil(size / cell); }

    // storage != nullptr: the grid adopts *storage (size nx * ny * nz, contents kept) instead of filling a fresh array
    // with default_value -- the SDF build seams hand over storage that the device-to-host drain writes exactly once.
    void Setup(const Eigen::Isometry3d& origin, const double cx, const double cy, const double cz,
               const int64_t nx, const int64_t ny, const int64_t nz, const T& default_value, const T& oob_value,
               BackingStore* storage = nullptr) {
        if (!(cx > 0.0) || !(cy > 0.0) || !(cz > 0.0)) throw std::invalid_argument("cell sizes must be positive");
        if (nx <= 0 || ny <= 0 || nz <= 0) throw std::invalid_argument("cell counts must be positive");
        if (storage && (int64_t)storage->size() != nx * ny * nz) throw std::invalid_argument("adopted storage has the wrong size");
        origin_transform_ = origin;
        inverse_origin_transform_ = origin.inverse();
        cell_x_size_ = cx; cell_y_size_ = cy; cell_z_size_ = cz;
        inv_cell_x_size_ = 1.0 / cx; inv_cell_y_size_ = 1.0 / cy; inv_cell_z_size_ = 1.0 / cz;
        num_x_cells_ = nx; num_y_cells_ = ny; num_z_cells_ = nz;
        x_size_ = (double)nx * cx; y_size_ = (double)ny * cy; z_size_ = (double)nz * cz;
        stride1_ = ny * nz;
        stride2_ = nz;
        default_value_ = default_value;
        oob_value_ = oob_value;
        if (storage) data_ = std::move(*storage);
        else data_.assign((size_t)(nx * ny * nz), default_value);
        initialized_ = true;
    }

    T& AccessIndex(const int64_t data_index) { return data_[(size_t)data_index]; }
    const T& AccessIndex(const int64_t data_index) const { return data_[(size_t)data_index]; }

public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW

    // cell-count constructors (the ones the SDF path uses: sdf_generation.hpp:105,245)
    VoxelGrid(const Eigen::Isometry3d& origin_transform, const double cell_size, const int64_t num_x_cells,
              const int64_t num_y_cells, const int64_t num_z_cells, const T& default_value) {
        Setup(origin_transform, cell_size, cell_size, cell_size, num_x_cells, num_y_cells, num_z_cells, default_value, default_value);
    }
    VoxelGrid(const Eigen::Isometry3d& origin_transform, const double cell_size, const int64_t num_x_cells,
              const int64_t num_y_cells, const int64_t num_z_cells, const T& default_value, const T& oob_value) {
        Setup(origin_transform, cell_size, cell_size, cell_size, num_x_cells, num_y_cells, num_z_cells, default_value, oob_value);
    }
    VoxelGrid(const double cell_size, const int64_t num_x_cells, const int64_t num_y_cells, const int64_t num_z_cells,
              const T& default_value) {
        Setup(Eigen::Isometry3d::Identity(), cell_size, cell_size, cell_size, num_x_cells, num_y_cells, num_z_cells, default_value, default_value);
    }
    VoxelGrid(const double cell_size, const int64_t num_x_cells, const int64_t num_y_cells, const int64_t num_z_cells,
              const T& default_value, const T& oob_value) {
        Setup(Eigen::Isometry3d::Identity(), cell_size, cell_size, cell_size, num_x_cells, num_y_cells, num_z_cells, default_value, oob_value);
    }
    // metric-size constructors (tutorial: 10 m at 0.25 m -> 40 cells)
    VoxelGrid(const Eigen::Isometry3d& origin_transform, const double cell_size, const double x_size,
              const double y_size, const double z_size, const T& default_value) {
        Setup(origin_transform, cell_size, cell_size, cell_size, CellsForSize(x_size, cell_size), CellsForSize(y_size, cell_size),
              CellsForSize(z_size, cell_size), default_value, default_value);
    }
    VoxelGrid(const Eigen::Isometry3d& origin_transform, const double cell_size, const double x_size,
              const double y_size, const double z_size, const T& default_value, const T& oob_value) {
        Setup(origin_transform, cell_size, cell_size, cell_size, CellsForSize(x_size, cell_size), CellsForSize(y_size, cell_size),
              CellsForSize(z_size, cell_size), default_value, oob_value);
    }
    VoxelGrid(const double cell_size, const double x_size, const double y_size, const double z_size, const T& default_value) {
        Setup(Eigen::Isometry3d::Identity(), cell_size, cell_size, cell_size, CellsForSize(x_size, cell_size),
              CellsForSize(y_size, cell_size), CellsForSize(z_size, cell_size), default_value, default_value);
    }
    VoxelGrid() {}
    virtual ~VoxelGrid() {}
    // (the user-provided destructor suppresses the implicit move operations: without these four lines every
    //  `std::move(grid)` -- the SDF seams return their 512 MiB result that way -- silently COPIES the array)
    VoxelGrid(const VoxelGrid&) = default;
    VoxelGrid(VoxelGrid&&) = default;
    VoxelGrid& operator=(const VoxelGrid&) = default;
    VoxelGrid& operator=(VoxelGrid&&) = default;

    virtual VoxelGrid<T, BackingStore>* Clone() const { return new VoxelGrid<T, BackingStore>(*this); }

    bool IsInitialized() const { return initialized_; }

    // ---- geometry -----------------------------------------------------------------------
    int64_t GetNumXCells() const { return num_x_cells_; }
    int64_t GetNumYCells() const { return num_y_cells_; }
    int64_t GetNumZCells() const { return num_z_cells_; }
    double GetXSize() const { return x_size_; }
    double GetYSize() const { return y_size_; }
    double GetZSize() const { return z_size_; }
    Eigen::Vector3d GetCellSizes() const { return Eigen::Vector3d(cell_x_size_, cell_y_size_, cell_z_size_); }
    const Eigen::Isometry3d& GetOriginTransform() const { return origin_transform_; }
    const Eigen::Isometry3d& GetInverseOriginTransform() const { return inverse_origin_transform_; }
    const T& GetDefaultValue() const { return default_value_; }
    const T& GetOOBValue() const { return oob_value_; }
    void SetOOBValue(const T& v) { oob_value_ = v; }

    bool IndexInBounds(const int64_t x, const int64_t y, const int64_t z) const {
        return x >= 0 && y >= 0 && z >= 0 && x < num_x_cells_ && y < num_y_cells_ && z < num_z_cells_;
    }
    bool IndexInBounds(const GRID_INDEX& i) const { return IndexInBounds(i.x, i.y, i.z); }
    int64_t GetDataIndex(const int64_t x, const int64_t y, const int64_t z) const { return x * stride1_ + y * stride2_ + z; }
    int64_t GetDataIndex(const GRID_INDEX& i) const { return GetDataIndex(i.x, i.y, i.z); }
    int64_t HashDataIndex(const int64_t x, const int64_t y, const int64_t z) const { return GetDataIndex(x, y, z); }

    GRID_INDEX PointInFrameToGridIndex4d(const Eigen::Vector4d& p) const {
        return GRID_INDEX((int64_t)std::floor(p(0) * inv_cell_x_size_), (int64_t)std::floor(p(1) * inv_cell_y_size_),
                          (int64_t)std::floor(p(2) * inv_cell_z_size_));
    }
    GRID_INDEX LocationToGridIndex4d(const Eigen::Vector4d& location) const {
        return PointInFrameToGridIndex4d(inverse_origin_transform_ * location);
    }
    GRID_INDEX LocationToGridIndex3d(const Eigen::Vector3d& location) const {
        const Eigen::Vector3d p = inverse_origin_transform_ * location;
        return PointInFrameToGridIndex4d(Eigen::Vector4d(p(0), p(1), p(2), 1.0));
    }
    GRID_INDEX LocationToGridIndex(const double x, const double y, const double z) const {
        return LocationToGridIndex4d(Eigen::Vector4d(x, y, z, 1.0));
    }
    bool LocationInBounds(const double x, const double y, const double z) const { return IndexInBounds(LocationToGridIndex(x, y, z)); }
    bool LocationInBounds4d(const Eigen::Vector4d& l) const { return IndexInBounds(LocationToGridIndex4d(l)); }

    Eigen::Vector4d GridIndexToLocationGridFrame(const int64_t x, const int64_t y, const int64_t z) const {
        return Eigen::Vector4d(cell_x_size_ * ((double)x + 0.5), cell_y_size_ * ((double)y + 0.5),
                               cell_z_size_ * ((double)z + 0.5), 1.0);
    }
    Eigen::Vector4d GridIndexToLocationGridFrame(const GRID_INDEX& i) const { return GridIndexToLocationGridFrame(i.x, i.y, i.z); }
    Eigen::Vector4d GridIndexToLocation(const int64_t x, const int64_t y, const int64_t z) const {
        return origin_transform_ * GridIndexToLocationGridFrame(x, y, z);
    }
    Eigen::Vector4d GridIndexToLocation(const GRID_INDEX& i) const { return GridIndexToLocation(i.x, i.y, i.z); }

    // ---- access: (value reference, in-bounds flag); out of bounds yields the OOB value ----
    std::pair<const T&, bool> GetImmutable(const int64_t x, const int64_t y, const int64_t z) const {
        if (IndexInBounds(x, y, z)) return std::pair<const T&, bool>(AccessIndex(GetDataIndex(x, y, z)), true);
        return std::pair<const T&, bool>(oob_value_, false);
    }
    std::pair<const T&, bool> GetImmutable(const GRID_INDEX& i) const { return GetImmutable(i.x, i.y, i.z); }
    std::pair<const T&, bool> GetImmutable4d(const Eigen::Vector4d& l) const { return GetImmutable(LocationToGridIndex4d(l)); }
    std::pair<const T&, bool> GetImmutable3d(const Eigen::Vector3d& l) const { return GetImmutable(LocationToGridIndex3d(l)); }
    std::pair<const T&, bool> GetImmutable(const double x, const double y, const double z) const {
        return GetImmutable(LocationToGridIndex(x, y, z));
    }
    virtual std::pair<T&, bool> GetMutable(const int64_t x, const int64_t y, const int64_t z) {
        if (IndexInBounds(x, y, z)) return std::pair<T&, bool>(AccessIndex(GetDataIndex(x, y, z)), true);
        return std::pair<T&, bool>(oob_value_, false);
    }
    virtual std::pair<T&, bool> GetMutable(const GRID_INDEX& i) { return GetMutable(i.x, i.y, i.z); }

    virtual bool SetValue(const int64_t x, const int64_t y, const int64_t z, const T& value) {
        if (!IndexInBounds(x, y, z)) return false;
        AccessIndex(GetDataIndex(x, y, z)) = value;
        return true;
    }
    virtual bool SetValue(const GRID_INDEX& i, const T& value) { return SetValue(i.x, i.y, i.z, value); }
    virtual bool SetValue4d(const Eigen::Vector4d& l, const T& value) { return SetValue(LocationToGridIndex4d(l), value); }
    virtual bool SetValue3d(const Eigen::Vector3d& l, const T& value) { return SetValue(LocationToGridIndex3d(l), value); }
    virtual bool SetValue(const double x, const double y, const double z, const T& value) {
        return SetValue(LocationToGridIndex(x, y, z), value);
    }

    const BackingStore& GetImmutableRawData() const { return data_; }
    BackingStore& GetMutableRawData() { return data_; }
    bool SetRawData(const BackingStore& data) {
        if ((int64_t)data.size() != num_x_cells_ * num_y_cells_ * num_z_cells_) return false;
        data_ = data;
        return true;
    }

    // ---- serialisation: field order of src/sdf_tools/sdf.cpp:221-253 --------------------------
    using ValueSerializer = std::function<uint64_t(const T&, std::vector<uint8_t>&)>;
    using ValueDeserializer = std::function<std::pair<T, uint64_t>(const std::vector<uint8_t>&, const uint64_t)>;

    uint64_t BaseSerializeSelf(std::vector<uint8_t>& buffer, const ValueSerializer& value_serializer) const {
        using namespace arc_utilities;
        const uint64_t start = buffer.size();
        SerializeFixedSizePOD<uint8_t>((uint8_t)initialized_, buffer);
        SerializeIsometry3d(origin_transform_, buffer);
        SerializeIsometry3d(inverse_origin_transform_, buffer);
        SerializeFixedSizePOD<uint64_t>((uint64_t)data_.size(), buffer);
        for (const T& v : data_) value_serializer(v, buffer);
        for (double d : {cell_x_size_, cell_y_size_, cell_z_size_, inv_cell_x_size_, inv_cell_y_size_, inv_cell_z_size_,
                         x_size_, y_size_, z_size_})
            SerializeFixedSizePOD<double>(d, buffer);
        for (int64_t i : {stride1_, stride2_, num_x_cells_, num_y_cells_, num_z_cells_}) SerializeFixedSizePOD<int64_t>(i, buffer);
        value_serializer(default_value_, buffer);
        value_serializer(oob_value_, buffer);
        return buffer.size() - start;
    }

    uint64_t BaseDeserializeSelf(const std::vector<uint8_t>& buffer, const uint64_t current, const ValueDeserializer& value_deserializer) {
        using namespace arc_utilities;
        uint64_t pos = current;
        const auto init = DeserializeFixedSizePOD<uint8_t>(buffer, pos); pos += init.second;
        const auto t0 = DeserializeIsometry3d(buffer, pos); pos += t0.second;
        const auto t1 = DeserializeIsometry3d(buffer, pos); pos += t1.second;
        const auto n = DeserializeFixedSizePOD<uint64_t>(buffer, pos); pos += n.second;
        // (untrusted bytes: every element takes at least one byte of the buffer, so a count beyond what is left is a
        //  corrupt or hostile header -- refuse it before reserving memory for it)
        if (pos > buffer.size() || n.first > (uint64_t)(buffer.size() - pos)) throw std::invalid_argument("serialized grid is inconsistent (element count exceeds the buffer)");
        BackingStore data;
        data.reserve((size_t)n.first);
        for (uint64_t i = 0; i < n.first; i++) { const auto v = value_deserializer(buffer, pos); data.push_back(v.first); pos += v.second; }
        double d[9];
        for (double& x : d) { const auto r = DeserializeFixedSizePOD<double>(buffer, pos); x = r.first; pos += r.second; }
        int64_t k[5];
        for (int64_t& x : k) { const auto r = DeserializeFixedSizePOD<int64_t>(buffer, pos); x = r.first; pos += r.second; }
        const auto dv = value_deserializer(buffer, pos); pos += dv.second;
        const auto ov = value_deserializer(buffer, pos); pos += ov.second;
        {   // dims >= 0 (the reference's default-constructed, uninitialised grid serialises 0 x 0 x 0), overflow-safe product
            typedef unsigned __int128 u128;
            const uint64_t sz = (uint64_t)data.size();
            bool ok = k[2] >= 0 && k[3] >= 0 && k[4] >= 0;
            if (ok && sz == 0) ok = k[2] == 0 || k[3] == 0 || k[4] == 0;
            else if (ok) {
                ok = (uint64_t)k[2] <= sz && (uint64_t)k[3] <= sz && (uint64_t)k[4] <= sz;
                const u128 p2 = (u128)(uint64_t)k[3] * (u128)(uint64_t)k[4];
                ok = ok && p2 <= (u128)sz && p2 * (u128)(uint64_t)k[2] == (u128)sz;
                ok = ok && k[0] == k[3] * k[4] && k[1] == k[4];                  // strides of the z-fastest layout
            }
            if (!ok) throw std::invalid_argument("serialized grid is inconsistent");
        }
        initialized_ = (bool)init.first;
        origin_transform_ = t0.first; inverse_origin_transform_ = t1.first;
        data_ = std::move(data);
        cell_x_size_ = d[0]; cell_y_size_ = d[1]; cell_z_size_ = d[2];
        inv_cell_x_size_ = d[3]; inv_cell_y_size_ = d[4]; inv_cell_z_size_ = d[5];
        x_size_ = d[6]; y_size_ = d[7]; z_size_ = d[8];
        stride1_ = k[0]; stride2_ = k[1]; num_x_cells_ = k[2]; num_y_cells_ = k[3]; num_z_cells_ = k[4];
        default_value_ = dv.first; oob_value_ = ov.first;
        return pos - current;
    }

    virtual uint64_t SerializeSelf(std::vector<uint8_t>& buffer, const ValueSerializer& value_serializer) const {
        return BaseSerializeSelf(buffer, value_serializer);
    }
    virtual uint64_t DeserializeSelf(const std::vector<uint8_t>& buffer, const uint64_t current,
                                     const ValueDeserializer& value_deserializer) {
        return BaseDeserializeSelf(buffer, current, value_deserializer);
    }
};

}  // namespace VoxelGrid
