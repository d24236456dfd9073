// TaggedObjectCollisionMapGrid -- the SDF-producing subset of sdf_tools::TaggedObjectCollisionMapGrid
// (reference include/sdf_tools/tagged_object_collision_map.hpp): the 16-byte cell (:22-44), the cell-count
// constructor, SetValue, and the four callers of the SDF hot path
//   ExtractFreeAndNamedObjectsSignedDistanceField (:730-811), ExtractSignedDistanceField(objects_to_use)
//   (:813-856), MakeObjectSDFs (:875-891), MakeAllObjectSDFs (:893-915).
// plus (round 4) its wire type: SerializeSelf / DeserializeSelf in the reference's field order
// (src/sdf_tools/tagged_object_collision_map.cpp:23-75, :77-240), TCMZ / TCMR files (:242-307) and the
// TaggedObjectCollisionMap message pair (:309-339, msg/TaggedObjectCollisionMap.msg) -- byte format of the un-vendored
// arc_utilities serialisers: "wire-format parity unpinned", like the other two containers.
// Connected components, convex segmentation, topology and RViz export are out of scope (SURVEY.md section 2,
// rows 7/8).  Every SDF is built on the GPU through sdfgpu_build_tagged_cells (device-side predicate).
#pragma once
#include <cstddef>
#include <cstdint>
#include <fstream>
#include <functional>
#include <iterator>
#include <limits>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "arc_utilities/serialization.hpp"
#include "arc_utilities/voxel_grid.hpp"
#include "arc_utilities/zlib_helpers.hpp"
#include "sdf_tools/sdf.hpp"
#include "sdf_tools/sdf_generation.hpp"

namespace sdf_tools {

// Plain mirror of msg/TaggedObjectCollisionMap.msg for builds without ROS (field names kept).
struct TaggedObjectCollisionMap {
    struct Header { uint32_t seq = 0; double stamp = 0.0; std::string frame_id; } header;
    std::vector<uint8_t> serialized_map;
    bool is_compressed = false;
};

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
        SignedDistanceField new_sdf(SignedDistanceField::ForBuild{}, GetOriginTransform(), frame_, cell_sizes.x(), GetNumXCells(),
                                    GetNumYCells(), GetNumZCells(), oob_value);
        double max_distance = 0.0, min_distance = 0.0;
        const std::shared_ptr<sdf_generation::SharedGpuContext> ctx = sdf_generation::GpuContext::Shared();
        const std::lock_guard<std::mutex> lock(ctx->mutex);
        sdfgpu_handle h = ctx->handle;
        sdf_generation::ThrowOnStatus(
            h, sdfgpu_build_tagged_cells(h, cells_on_device ? nullptr : data_.data(), sizeof(TAGGED_OBJECT_COLLISION_CELL),
                                         offsetof(TAGGED_OBJECT_COLLISION_CELL, occupancy),
                                         offsetof(TAGGED_OBJECT_COLLISION_CELL, object_id), object_mode,
                                         ids.empty() ? nullptr : ids.data(), (int64_t)ids.size(), unknown_is_filled ? 1 : 0,
                                         GetNumXCells(), GetNumYCells(), GetNumZCells(), cell_sizes.x(),
                                         add_virtual_border ? 1 : 0, new_sdf.MutableDataForBuild(), &max_distance,
                                         &min_distance));
        return std::make_pair(std::move(new_sdf), std::make_pair(max_distance, min_distance));      // (moved: a copy of the field costs as much as its download)
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

    // ---- wire formats: tagged_object_collision_map.cpp:23-75 (fields), :242-307 (files), :309-339 (messages) ------------
    using CellSerializer = std::function<uint64_t(const TAGGED_OBJECT_COLLISION_CELL&, std::vector<uint8_t>&)>;
    using CellDeserializer = std::function<std::pair<TAGGED_OBJECT_COLLISION_CELL, uint64_t>(const std::vector<uint8_t>&, const uint64_t)>;

    uint64_t SerializeSelf(std::vector<uint8_t>& buffer,
                           const CellSerializer& value_serializer = arc_utilities::SerializeFixedSizePOD<TAGGED_OBJECT_COLLISION_CELL>) const override {
        (void)value_serializer;                                   // (ignored by the reference too: cells are fixed-size PODs, :29)
        const uint64_t start = buffer.size();
        BaseSerializeSelf(buffer, arc_utilities::SerializeFixedSizePOD<TAGGED_OBJECT_COLLISION_CELL>);   // initialized .. OOB value (:31-63)
        arc_utilities::SerializeFixedSizePOD<uint32_t>(number_of_components_, buffer);                   // (:65)
        arc_utilities::SerializeFixedSizePOD<uint32_t>(number_of_convex_segments_, buffer);              // (:66)
        arc_utilities::SerializeString(frame_, buffer);                                                  // (:67)
        arc_utilities::SerializeFixedSizePOD<uint8_t>((uint8_t)components_valid_, buffer);               // (:68)
        arc_utilities::SerializeFixedSizePOD<uint8_t>((uint8_t)convex_segments_valid_, buffer);          // (:70)
        return buffer.size() - start;
    }
    uint64_t DeserializeSelf(const std::vector<uint8_t>& buffer, const uint64_t current,
                             const CellDeserializer& value_deserializer = arc_utilities::DeserializeFixedSizePOD<TAGGED_OBJECT_COLLISION_CELL>) override {
        (void)value_deserializer;
        uint64_t pos = current;
        pos += BaseDeserializeSelf(buffer, pos, arc_utilities::DeserializeFixedSizePOD<TAGGED_OBJECT_COLLISION_CELL>);
        const auto nc = arc_utilities::DeserializeFixedSizePOD<uint32_t>(buffer, pos); pos += nc.second;
        const auto ns = arc_utilities::DeserializeFixedSizePOD<uint32_t>(buffer, pos); pos += ns.second;
        const auto fr = arc_utilities::DeserializeString(buffer, pos); pos += fr.second;
        const auto cv = arc_utilities::DeserializeFixedSizePOD<uint8_t>(buffer, pos); pos += cv.second;
        const auto sv = arc_utilities::DeserializeFixedSizePOD<uint8_t>(buffer, pos); pos += sv.second;
        number_of_components_ = nc.first;
        number_of_convex_segments_ = ns.first;
        frame_ = fr.first;
        components_valid_ = (bool)cv.first;
        convex_segments_valid_ = (bool)sv.first;
        return pos - current;
    }

    static void SaveToFile(const TaggedObjectCollisionMapGrid& map, const std::string& filepath, const bool compress) {
        std::vector<uint8_t> buffer;
        map.SerializeSelf(buffer);
        std::ofstream out(filepath, std::ios::out | std::ios::binary);
        const std::vector<uint8_t> body = compress ? ZlibHelpers::CompressBytes(buffer) : buffer;
        out.write(compress ? "TCMZ" : "TCMR", 4);                 // 4-byte magic (:251-263)
        out.write(reinterpret_cast<const char*>(body.data()), (std::streamsize)body.size());
    }
    static TaggedObjectCollisionMapGrid LoadFromFile(const std::string& filepath) {
        std::ifstream in(filepath, std::ios::in | std::ios::binary);
        if (!in.good()) throw std::invalid_argument("File does not exist");
        std::vector<uint8_t> all((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        if (all.size() < 4) throw std::invalid_argument("File is too small");
        const std::string magic(all.begin(), all.begin() + 4);
        const std::vector<uint8_t> body(all.begin() + 4, all.end());
        TaggedObjectCollisionMapGrid map;
        if (magic == "TCMZ") map.DeserializeSelf(ZlibHelpers::DecompressBytes(body), 0);
        else if (magic == "TCMR") map.DeserializeSelf(body, 0);
        else throw std::invalid_argument("File has invalid header [" + magic + "]");
        return map;
    }
    static TaggedObjectCollisionMap GetMessageRepresentation(const TaggedObjectCollisionMapGrid& map) {
        TaggedObjectCollisionMap msg;                             // always zlib-compressed (:312-321); no ROS clock here: stamp stays 0
        msg.header.frame_id = map.GetFrame();
        std::vector<uint8_t> buffer;
        map.SerializeSelf(buffer);
        msg.serialized_map = ZlibHelpers::CompressBytes(buffer);
        msg.is_compressed = true;
        return msg;
    }
    static TaggedObjectCollisionMapGrid LoadFromMessageRepresentation(const TaggedObjectCollisionMap& message) {
        TaggedObjectCollisionMapGrid map;
        if (message.is_compressed) map.DeserializeSelf(ZlibHelpers::DecompressBytes(message.serialized_map), 0);
        else map.DeserializeSelf(message.serialized_map, 0);
        return map;
    }

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
