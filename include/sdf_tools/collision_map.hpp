// CollisionMapGrid -- dense occupancy grid, API-compatible with the subset of
// sdf_tools::CollisionMapGrid (reference include/sdf_tools/collision_map.hpp) on the SDF path:
// COLLISION_CELL (:20-32), the constructors (:215-270), SetValue (:405-420) and
// ExtractSignedDistanceField (:680-712).  Connected components, topology and RViz export are out
// of scope.  Wire formats (N3): SerializeSelf / DeserializeSelf, SaveToFile /
// LoadFromFile ("CMGZ" / "CMGR") and the CollisionMap message pair in the field order of
// src/sdf_tools/collision_map.cpp:21-62, :205-283, :285-315.  The byte layout of the primitives
// (arc_utilities::SerializeFixedSizePOD / SerializeEigen / SerializeVector / SerializeString) is the in-tree
// include/arc_utilities/serialization.hpp: arc_utilities is not vendored in the reference checkout, so byte-level
// interoperability with files written by the reference is UNVERIFIED.
#pragma once
#include <cstddef>
#include <cstdint>
#include <fstream>
#include <functional>
#include <iterator>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "arc_utilities/serialization.hpp"
#include "arc_utilities/voxel_grid.hpp"
#include "arc_utilities/zlib_helpers.hpp"
#include "sdf_tools/sdf.hpp"
#include "sdf_tools/sdf_generation.hpp"

namespace sdf_tools {

struct COLLISION_CELL {
    float occupancy;
    uint32_t component;
    COLLISION_CELL() : occupancy(0.0), component(0) {}
    explicit COLLISION_CELL(const float in_occupancy) : occupancy(in_occupancy), component(0) {}
    COLLISION_CELL(const float in_occupancy, const uint32_t in_component) : occupancy(in_occupancy), component(in_component) {}
};
static_assert(sizeof(COLLISION_CELL) == 8, "COLLISION_CELL must stay an 8-byte record (device classify kernel)");

// Plain mirror of msg/CollisionMap.msg for builds without ROS (field names kept).
struct CollisionMap {
    struct Header { uint32_t seq = 0; double stamp = 0.0; std::string frame_id; } header;
    std::vector<uint8_t> serialized_map;
    bool is_compressed = false;
};

class CollisionMapGrid : public VoxelGrid::VoxelGrid<COLLISION_CELL> {
protected:
    uint32_t number_of_components_;
    std::string frame_;
    bool components_valid_;

public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    using Base = ::VoxelGrid::VoxelGrid<COLLISION_CELL>;

    CollisionMapGrid(const Eigen::Isometry3d& origin_transform, const std::string& frame, const double resolution,
                     const int64_t x_cells, const int64_t y_cells, const int64_t z_cells, const COLLISION_CELL& oob_default_value)
        : Base(origin_transform, resolution, x_cells, y_cells, z_cells, oob_default_value), number_of_components_(0u), frame_(frame), components_valid_(false) {}
    CollisionMapGrid(const std::string& frame, const double resolution, const int64_t x_cells, const int64_t y_cells,
                     const int64_t z_cells, const COLLISION_CELL& oob_default_value)
        : Base(resolution, x_cells, y_cells, z_cells, oob_default_value), number_of_components_(0u), frame_(frame), components_valid_(false) {}
    CollisionMapGrid(const Eigen::Isometry3d& origin_transform, const std::string& frame, const double resolution,
                     const int64_t x_cells, const int64_t y_cells, const int64_t z_cells, const COLLISION_CELL& default_value,
                     const COLLISION_CELL& OOB_value)
        : Base(origin_transform, resolution, x_cells, y_cells, z_cells, default_value, OOB_value), number_of_components_(0u), frame_(frame), components_valid_(false) {}
    CollisionMapGrid(const Eigen::Isometry3d& origin_transform, const std::string& frame, const double resolution,
                     const double x_size, const double y_size, const double z_size, const COLLISION_CELL& oob_default_value)
        : Base(origin_transform, resolution, x_size, y_size, z_size, oob_default_value), number_of_components_(0u), frame_(frame), components_valid_(false) {}
    CollisionMapGrid() : Base(), number_of_components_(0u), frame_(""), components_valid_(false) {}

    Base* Clone() const override { return new CollisionMapGrid(*this); }

    double GetResolution() const { return GetCellSizes().x(); }
    std::string GetFrame() const { return frame_; }
    void SetFrame(const std::string& f) { frame_ = f; }
    bool AreComponentsValid() const { return components_valid_; }

    bool SetValue(const int64_t x, const int64_t y, const int64_t z, const COLLISION_CELL& value) override {
        if (!IndexInBounds(x, y, z)) return false;
        components_valid_ = false;
        AccessIndex(GetDataIndex(x, y, z)) = value;
        return true;
    }
    bool SetValue(const GRID_INDEX& i, const COLLISION_CELL& v) override { return SetValue(i.x, i.y, i.z, v); }
    bool SetValue4d(const Eigen::Vector4d& l, const COLLISION_CELL& v) override { return SetValue(LocationToGridIndex4d(l), v); }
    bool SetValue3d(const Eigen::Vector3d& l, const COLLISION_CELL& v) override { return SetValue(LocationToGridIndex3d(l), v); }
    bool SetValue(const double x, const double y, const double z, const COLLISION_CELL& v) override {
        return SetValue(LocationToGridIndex(x, y, z), v);
    }

    // occupancy > 0.5 is filled; == 0.5 is "unknown" and filled only on request (reference :689-704).
    // Every cell is in bounds by construction, so the reference's "index out of grid bounds" throw
    // (:707) cannot trigger; the raw cell array is classified on the device.
    std::pair<SignedDistanceField, std::pair<double, double>> ExtractSignedDistanceField(
        const float oob_value, const bool unknown_is_filled, const bool add_virtual_border) const {
        return sdf_generation::ExtractSignedDistanceFieldFromCells(
            GetOriginTransform(), GetCellSizes(), GetNumXCells(), GetNumYCells(), GetNumZCells(), data_.data(),
            sizeof(COLLISION_CELL), offsetof(COLLISION_CELL, occupancy), unknown_is_filled, oob_value, GetFrame(), add_virtual_border);
    }

    // The same build with the field left in HBM (round 4): batched EstimateDistance / GetGradient queries run on the
    // device, Host() downloads the reference's container lazily (include/sdf_tools/device_sdf.hpp).
    std::pair<DeviceSignedDistanceField, std::pair<double, double>> ExtractSignedDistanceFieldDevice(
        const float oob_value, const bool unknown_is_filled, const bool add_virtual_border) const {
        return sdf_generation::ExtractSignedDistanceFieldDeviceFromCells(
            GetOriginTransform(), GetCellSizes(), GetNumXCells(), GetNumYCells(), GetNumZCells(), data_.data(),
            sizeof(COLLISION_CELL), offsetof(COLLISION_CELL, occupancy), unknown_is_filled, oob_value, GetFrame(), add_virtual_border);
    }

    // ---- wire formats: collision_map.cpp:21-62 (fields), :205-283 (files), :285-315 (messages) ------------------------
    using CellSerializer = std::function<uint64_t(const COLLISION_CELL&, std::vector<uint8_t>&)>;
    using CellDeserializer = std::function<std::pair<COLLISION_CELL, uint64_t>(const std::vector<uint8_t>&, const uint64_t)>;

    uint64_t SerializeSelf(std::vector<uint8_t>& buffer,
                           const CellSerializer& value_serializer = arc_utilities::SerializeFixedSizePOD<COLLISION_CELL>) const override {
        (void)value_serializer;                                   // (the reference ignores it too: cells are fixed-size PODs)
        const uint64_t start = buffer.size();
        BaseSerializeSelf(buffer, arc_utilities::SerializeFixedSizePOD<COLLISION_CELL>);   // initialized .. OOB value (:28-57)
        arc_utilities::SerializeFixedSizePOD<uint32_t>(number_of_components_, buffer);     // (:59)
        arc_utilities::SerializeString(frame_, buffer);                                    // (:60)
        arc_utilities::SerializeFixedSizePOD<uint8_t>((uint8_t)components_valid_, buffer); // (:61)
        return buffer.size() - start;
    }
    uint64_t DeserializeSelf(const std::vector<uint8_t>& buffer, const uint64_t current,
                             const CellDeserializer& value_deserializer = arc_utilities::DeserializeFixedSizePOD<COLLISION_CELL>) override {
        (void)value_deserializer;
        uint64_t pos = current;
        pos += BaseDeserializeSelf(buffer, pos, arc_utilities::DeserializeFixedSizePOD<COLLISION_CELL>);
        const auto nc = arc_utilities::DeserializeFixedSizePOD<uint32_t>(buffer, pos); pos += nc.second;
        const auto fr = arc_utilities::DeserializeString(buffer, pos); pos += fr.second;
        const auto cv = arc_utilities::DeserializeFixedSizePOD<uint8_t>(buffer, pos); pos += cv.second;
        number_of_components_ = nc.first;
        frame_ = fr.first;
        components_valid_ = (bool)cv.first;
        return pos - current;
    }

    static void SaveToFile(const CollisionMapGrid& map, const std::string& filepath, const bool compress) {
        std::vector<uint8_t> buffer;
        map.SerializeSelf(buffer);
        std::ofstream out(filepath, std::ios::out | std::ios::binary);
        const std::vector<uint8_t> body = compress ? ZlibHelpers::CompressBytes(buffer) : buffer;
        out.write(compress ? "CMGZ" : "CMGR", 4);                 // 4-byte magic (:214-229)
        out.write(reinterpret_cast<const char*>(body.data()), (std::streamsize)body.size());
    }
    static CollisionMapGrid LoadFromFile(const std::string& filepath) {
        std::ifstream in(filepath, std::ios::in | std::ios::binary);
        if (!in.good()) throw std::invalid_argument("File does not exist");
        std::vector<uint8_t> all((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        if (all.size() < 4) throw std::invalid_argument("File is too small");
        const std::string magic(all.begin(), all.begin() + 4);
        const std::vector<uint8_t> body(all.begin() + 4, all.end());
        CollisionMapGrid map;
        if (magic == "CMGZ") map.DeserializeSelf(ZlibHelpers::DecompressBytes(body), 0);
        else if (magic == "CMGR") map.DeserializeSelf(body, 0);
        else throw std::invalid_argument("File has invalid header [" + magic + "]");
        return map;
    }
    static CollisionMap GetMessageRepresentation(const CollisionMapGrid& map) {
        CollisionMap msg;                                         // always zlib-compressed (:285-296); no ROS clock here: stamp stays 0
        msg.header.frame_id = map.GetFrame();
        std::vector<uint8_t> buffer;
        map.SerializeSelf(buffer);
        msg.serialized_map = ZlibHelpers::CompressBytes(buffer);
        msg.is_compressed = true;
        return msg;
    }
    static CollisionMapGrid LoadFromMessageRepresentation(const CollisionMap& message) {
        CollisionMapGrid map;
        if (message.is_compressed) map.DeserializeSelf(ZlibHelpers::DecompressBytes(message.serialized_map), 0);
        else map.DeserializeSelf(message.serialized_map, 0);
        return map;
    }

    // Same result through the generic predicate seam (kept for callers that pass their own predicate).
    std::pair<SignedDistanceField, std::pair<double, double>> ExtractSignedDistanceFieldViaPredicate(
        const float oob_value, const bool unknown_is_filled, const bool add_virtual_border) const {
        const std::function<bool(const GRID_INDEX&)> is_filled_fn = [&](const GRID_INDEX& index) {
            const auto query = GetImmutable(index);
            if (!query.second) throw std::runtime_error("index out of grid bounds");
            return (query.first.occupancy > 0.5) || (unknown_is_filled && (query.first.occupancy == 0.5));
        };
        return sdf_generation::ExtractSignedDistanceField(*this, is_filled_fn, oob_value, GetFrame(), add_virtual_border);
    }
};

}  // namespace sdf_tools
