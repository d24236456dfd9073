// CollisionMapGrid -- dense occupancy grid, API-compatible with the subset of
// sdf_tools::CollisionMapGrid (reference include/sdf_tools/collision_map.hpp) on the SDF path:
// COLLISION_CELL (:20-32), the constructors (:215-270), SetValue (:405-420) and
// ExtractSignedDistanceField (:680-712).  Connected components, topology and RViz export are out
// of scope.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "arc_utilities/voxel_grid.hpp"
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
