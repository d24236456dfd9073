// TaggedObjectCollisionMapGrid -- the SDF-producing subset of sdf_tools::TaggedObjectCollisionMapGrid
// (reference include/sdf_tools/tagged_object_collision_map.hpp): the 16-byte cell (:22-44), the cell-count
// constructor, SetValue, and the four callers of the SDF hot path
//   ExtractFreeAndNamedObjectsSignedDistanceField (:730-811), ExtractSignedDistanceField(objects_to_use)
//   (:813-856), MakeObjectSDFs (:875-891), MakeAllObjectSDFs (:893-915).
// Connected components, convex segmentation, topology and RViz export are out of scope (SURVEY.md section 2,
// rows 7/8).  Every SDF is built on the GPU through sdfgpu_build_tagged_cells (device-side predicate).
#pragma once
#include <cstddef>
#include <cstdint>
#include <limits>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "arc_utilities/voxel_grid.hpp"
#include "sdf_tools/sdf.hpp"
#include "sdf_tools/sdf_generation.hpp"

namespace sdf_tools {

struct TAGGED_OBJECT_COLLISION_CELL {
    float occupancy;
    uint32_t component;
    uint32_t object_id;
    uint32_t convex_segment;
    TAGGED_OBJECT_COLLISION_CELL() : occupancy(0.0), component(0u), object_id(0u), convex_segment(0u) {}
    TAGGED_OBJECT_COLLISION_CELL(const float in_occupancy, const uint32_t in_object_id)
        : occupancy(in_occupancy), component(0u), object_id(in_object_id), convex_segment(0u) {}
    TAGGED_OBJECT_COLLISION_CELL(const float in_occupancy, const uint32_t in_object_id, const uint32_t in_component,
                                 const uint32_t in_convex_segment)
        : occupancy(in_occupancy), component(in_component), object_id(in_object_id), convex_segment(in_convex_segment) {}
};
static_assert(sizeof(TAGGED_OBJECT_COLLISION_CELL) == 16, "TAGGED_OBJECT_COLLISION_CELL must stay a 16-byte record");

class TaggedObjectCollisionMapGrid : public VoxelGrid::VoxelGrid<TAGGED_OBJECT_COLLISION_CELL> {
protected:
    uint32_t number_of_components_;
    uint32_t number_of_convex_segments_;
    std::string frame_;
    bool components_valid_;
    bool convex_segments_valid_;

    std::pair<SignedDistanceField, std::pair<double, double>> BuildWithFilter(
        const float oob_value, const int object_mode, const std::vector<uint32_t>& ids, const bool unknown_is_filled,
        const bool add_virtual_border, const bool cells_on_device = false) const {
        const Eigen::Vector3d cell_sizes = GetCellSizes();
        if ((cell_sizes.x() != cell_sizes.y()) || (cell_sizes.x() != cell_sizes.z()))
            throw std::invalid_argument("Grid must have uniform resolution");
        SignedDistanceField new_sdf(GetOriginTransform(), frame_, cell_sizes.x(), GetNumXCells(), GetNumYCells(),
                                    GetNumZCells(), oob_value);
        double max_distance = 0.0, min_distance = 0.0;
        sdfgpu_handle h = sdf_generation::GpuContext::Get();
        sdf_generation::ThrowOnStatus(
            h, sdfgpu_build_tagged_cells(h, cells_on_device ? nullptr : data_.data(), sizeof(TAGGED_OBJECT_COLLISION_CELL),
                                         offsetof(TAGGED_OBJECT_COLLISION_CELL, occupancy),
                                         offsetof(TAGGED_OBJECT_COLLISION_CELL, object_id), object_mode,
                                         ids.empty() ? nullptr : ids.data(), (int64_t)ids.size(), unknown_is_filled ? 1 : 0,
                                         GetNumXCells(), GetNumYCells(), GetNumZCells(), cell_sizes.x(),
                                         add_virtual_border ? 1 : 0, new_sdf.MutableDataForBuild(), &max_distance,
                                         &min_distance));
        return std::make_pair(new_sdf, std::make_pair(max_distance, min_distance));
    }

public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    using Base = ::VoxelGrid::VoxelGrid<TAGGED_OBJECT_COLLISION_CELL>;

    TaggedObjectCollisionMapGrid(const Eigen::Isometry3d& origin_transform, const std::string& frame, const double resolution,
                                 const int64_t x_cells, const int64_t y_cells, const int64_t z_cells,
                                 const TAGGED_OBJECT_COLLISION_CELL& oob_default_value)
        : Base(origin_transform, resolution, x_cells, y_cells, z_cells, oob_default_value), number_of_components_(0u),
          number_of_convex_segments_(0u), frame_(frame), components_valid_(false), convex_segments_valid_(false) {}
    TaggedObjectCollisionMapGrid(const std::string& frame, const double resolution, const int64_t x_cells, const int64_t y_cells,
                                 const int64_t z_cells, const TAGGED_OBJECT_COLLISION_CELL& oob_default_value)
        : Base(resolution, x_cells, y_cells, z_cells, oob_default_value), number_of_components_(0u),
          number_of_convex_segments_(0u), frame_(frame), components_valid_(false), convex_segments_valid_(false) {}
    TaggedObjectCollisionMapGrid()
        : Base(), number_of_components_(0u), number_of_convex_segments_(0u), frame_(""), components_valid_(false),
          convex_segments_valid_(false) {}

    Base* Clone() const override { return new TaggedObjectCollisionMapGrid(*this); }
    double GetResolution() const { return GetCellSizes().x(); }
    std::string GetFrame() const { return frame_; }
    void SetFrame(const std::string& f) { frame_ = f; }

    bool SetValue(const int64_t x, const int64_t y, const int64_t z, const TAGGED_OBJECT_COLLISION_CELL& value) override {
        if (!IndexInBounds(x, y, z)) return false;
        components_valid_ = false;
        convex_segments_valid_ = false;
        AccessIndex(GetDataIndex(x, y, z)) = value;
        return true;
    }
    bool SetValue(const GRID_INDEX& i, const TAGGED_OBJECT_COLLISION_CELL& v) override { return SetValue(i.x, i.y, i.z, v); }

    // Filled = occupied cell whose object id is in objects_to_use (any object if the list is empty), :813-856.
    std::pair<SignedDistanceField, std::pair<double, double>> ExtractSignedDistanceField(
        const float oob_value, const std::vector<uint32_t>& objects_to_use, const bool unknown_is_filled,
        const bool add_virtual_border) const {
        return BuildWithFilter(oob_value, 2, objects_to_use, unknown_is_filled, add_virtual_border);
    }

    // Free-space SDF outside, named-object SDF inside, 0 in filled cells that belong to no named object, :730-811.
    std::pair<SignedDistanceField, std::pair<double, double>> ExtractFreeAndNamedObjectsSignedDistanceField(
        const float oob_value, const bool unknown_is_filled) const {
        const auto free_sdf_result = BuildWithFilter(oob_value, 0, {}, unknown_is_filled, false);
        // (same records as the call above: they are still on the device)
        const auto named_objects_sdf_result = BuildWithFilter(oob_value, 1, {}, unknown_is_filled, false, true);
        SignedDistanceField combined_sdf = free_sdf_result.first;
        const std::vector<float>& fr = free_sdf_result.first.GetImmutableRawData();
        const std::vector<float>& nm = named_objects_sdf_result.first.GetImmutableRawData();
        float* out = combined_sdf.MutableDataForBuild();
        for (size_t i = 0; i < fr.size(); i++) {
            if (fr[i] >= 0.0) out[i] = fr[i];
            else if (nm[i] <= -0.0) out[i] = nm[i];
            else out[i] = 0.0f;
        }
        return std::make_pair(combined_sdf, std::make_pair(free_sdf_result.second.first, named_objects_sdf_result.second.second));
    }

    // One SDF per object id, built with oob = +inf like the reference (:875-891).
    std::map<uint32_t, SignedDistanceField> MakeObjectSDFs(const std::vector<uint32_t>& object_ids, const bool unknown_is_filled,
                                                           const bool add_virtual_border) const {
        std::map<uint32_t, SignedDistanceField> per_object_sdfs;
        bool uploaded = false;                  // the cell records travel once; every further object re-uses the device copy
        for (const uint32_t object_id : object_ids) {
            per_object_sdfs[object_id] = BuildWithFilter(std::numeric_limits<float>::infinity(), 2,
                                                         std::vector<uint32_t>{object_id}, unknown_is_filled,
                                                         add_virtual_border, uploaded).first;
            uploaded = true;
        }
        return per_object_sdfs;
    }

    // ... for every object id > 0 present in the grid (:893-915).
    std::map<uint32_t, SignedDistanceField> MakeAllObjectSDFs(const bool unknown_is_filled, const bool add_virtual_border) const {
        std::map<uint32_t, uint32_t> object_id_map;
        for (const TAGGED_OBJECT_COLLISION_CELL& cell : data_)
            if (cell.object_id > 0) object_id_map[cell.object_id] = 1u;
        std::vector<uint32_t> ids;
        for (const auto& kv : object_id_map) ids.push_back(kv.first);
        return MakeObjectSDFs(ids, unknown_is_filled, add_virtual_border);
    }
};

}  // namespace sdf_tools
