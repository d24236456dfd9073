// pysdf_tools -- pybind11 module with the surface of the reference's src/sdf_tools/bindings.cpp
// (module name, class names, method names and positional argument orders, :15-106), bound to the
// in-tree mirror classes whose SDF build runs on the MI355X through the C ABI (libsdfgpu.so).
// Extras beside (not instead of) the reference-shaped methods: numpy fast paths
// (SetOccupancyFromNumpy / GetRawDataNumpy / GetFullGradientNumpy) that avoid per-voxel Python calls,
// and the GIL is released for the duration of ExtractSignedDistanceField.
#include <pybind11/functional.h>
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstdint>
#include <cstring>

#include "arc_utilities/zlib_helpers.hpp"
#include "sdf_tools/collision_map.hpp"
#include "sdf_tools/device_sdf.hpp"
#include "sdf_tools/tagged_object_collision_map.hpp"

namespace py = pybind11;
using namespace sdf_tools;
using Eigen::Isometry3d;

namespace {

Isometry3d IsometryFromArray(const py::array_t<double, py::array::c_style | py::array::forcecast>& m) {
    if (m.ndim() != 2 || m.shape(0) != 4 || m.shape(1) != 4) throw std::invalid_argument("Isometry3d expects a 4x4 matrix");
    Isometry3d t = Isometry3d::Identity();
    auto a = m.unchecked<2>();
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) t.matrix()(r, c) = a(r, c);
    return t;
}

py::array_t<double> MatrixToArray(const Isometry3d& t) {
    py::array_t<double> out({4, 4});
    auto a = out.mutable_unchecked<2>();
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) a(r, c) = t.matrix()(r, c);
    return out;
}

}  // namespace

PYBIND11_MODULE(pysdf_tools, m) {
    m.doc() = "MI355X-native drop-in for sdf_tools' pysdf_tools (SDF build on the GPU via libsdfgpu.so)";

    py::class_<COLLISION_CELL>(m, "COLLISION_CELL")
        .def(py::init<float>())
        .def(py::init<float, uint32_t>())
        .def_readwrite("occupancy", &COLLISION_CELL::occupancy)
        .def_readwrite("component", &COLLISION_CELL::component);

    py::class_<Isometry3d>(m, "Isometry3d")
        .def(py::init(&IsometryFromArray))
        .def("translation", [](const Isometry3d& t) {
            const auto v = t.translation();
            py::array_t<double> out(3);
            out.mutable_at(0) = v(0); out.mutable_at(1) = v(1); out.mutable_at(2) = v(2);
            return out;
        })
        .def("matrix", &MatrixToArray);

    py::class_<SDF>(m, "SDF")
        .def(py::init<>())
        .def_readwrite("serialized_sdf", &SDF::serialized_sdf)
        .def_readwrite("is_compressed", &SDF::is_compressed)
        .def_property("frame_id", [](const SDF& s) { return s.header.frame_id; }, [](SDF& s, const std::string& f) { s.header.frame_id = f; });

    py::class_<CollisionMap>(m, "CollisionMap")                // plain mirror of msg/CollisionMap.msg
        .def(py::init<>())
        .def_readwrite("serialized_map", &CollisionMap::serialized_map)
        .def_readwrite("is_compressed", &CollisionMap::is_compressed)
        .def_property("frame_id", [](const CollisionMap& s) { return s.header.frame_id; }, [](CollisionMap& s, const std::string& f) { s.header.frame_id = f; });

    // The tagged-object map (not in the reference's pybind surface, bindings.cpp:15-106; bound here beside it so that its wire
    // type and its SDF callers -- tagged_object_collision_map.hpp:730-915, .cpp:23-339 -- can be driven from Python tests)
    py::class_<TAGGED_OBJECT_COLLISION_CELL>(m, "TAGGED_OBJECT_COLLISION_CELL")
        .def(py::init<>())
        .def(py::init<float, uint32_t>())
        .def(py::init<float, uint32_t, uint32_t, uint32_t>())
        .def_readwrite("occupancy", &TAGGED_OBJECT_COLLISION_CELL::occupancy)
        .def_readwrite("component", &TAGGED_OBJECT_COLLISION_CELL::component)
        .def_readwrite("object_id", &TAGGED_OBJECT_COLLISION_CELL::object_id)
        .def_readwrite("convex_segment", &TAGGED_OBJECT_COLLISION_CELL::convex_segment);
    py::class_<TaggedObjectCollisionMap>(m, "TaggedObjectCollisionMap")   // plain mirror of msg/TaggedObjectCollisionMap.msg
        .def(py::init<>())
        .def_readwrite("serialized_map", &TaggedObjectCollisionMap::serialized_map)
        .def_readwrite("is_compressed", &TaggedObjectCollisionMap::is_compressed)
        .def_property("frame_id", [](const TaggedObjectCollisionMap& s) { return s.header.frame_id; },
                      [](TaggedObjectCollisionMap& s, const std::string& f) { s.header.frame_id = f; });
    py::class_<TaggedObjectCollisionMapGrid>(m, "TaggedObjectCollisionMapGrid")
        .def(py::init<Isometry3d const&, std::string, double, int64_t, int64_t, int64_t, TAGGED_OBJECT_COLLISION_CELL const&>())
        .def(py::init<>())
        .def("SetValue", [](TaggedObjectCollisionMapGrid& g, int64_t x, int64_t y, int64_t z, const TAGGED_OBJECT_COLLISION_CELL& c) { return g.SetValue(x, y, z, c); })
        .def("GetValueByIndex", [](const TaggedObjectCollisionMapGrid& g, int64_t x, int64_t y, int64_t z) { const auto q = g.GetImmutable(x, y, z); return std::make_pair(q.first, q.second); })
        .def("GetNumXCells", &TaggedObjectCollisionMapGrid::GetNumXCells)
        .def("GetNumYCells", &TaggedObjectCollisionMapGrid::GetNumYCells)
        .def("GetNumZCells", &TaggedObjectCollisionMapGrid::GetNumZCells)
        .def("GetFrame", &TaggedObjectCollisionMapGrid::GetFrame)
        .def("GetResolution", &TaggedObjectCollisionMapGrid::GetResolution)
        .def("SerializeSelf", [](const TaggedObjectCollisionMapGrid& g) { std::vector<uint8_t> b; g.SerializeSelf(b); return py::bytes(reinterpret_cast<const char*>(b.data()), b.size()); })
        .def_static("Deserialize", [](const py::bytes& data) { const std::string s = data; TaggedObjectCollisionMapGrid g; g.DeserializeSelf(std::vector<uint8_t>(s.begin(), s.end()), 0); return g; })
        .def("SaveToFile", [](const TaggedObjectCollisionMapGrid& g, const std::string& path, bool compress) { TaggedObjectCollisionMapGrid::SaveToFile(g, path, compress); })
        .def_static("LoadFromFile", &TaggedObjectCollisionMapGrid::LoadFromFile)
        .def("GetMessageRepresentation", [](const TaggedObjectCollisionMapGrid& g) { return TaggedObjectCollisionMapGrid::GetMessageRepresentation(g); })
        .def_static("LoadFromMessageRepresentation", &TaggedObjectCollisionMapGrid::LoadFromMessageRepresentation)
        .def("ExtractSignedDistanceField", [](const TaggedObjectCollisionMapGrid& g, float oob_value, const std::vector<uint32_t>& objects_to_use,
                                               bool unknown_is_filled, bool add_virtual_border) {
            return g.ExtractSignedDistanceField(oob_value, objects_to_use, unknown_is_filled, add_virtual_border);
        }, py::call_guard<py::gil_scoped_release>());

    using VoxelGridVecd = VoxelGrid::VoxelGrid<std::vector<double>>;

    py::class_<SignedDistanceField>(m, "SignedDistanceField")
        .def(py::init<>())
        .def("GetRawData", &SignedDistanceField::GetImmutableRawData, "Please don't mutate this")
        .def("GetFullGradient", &SignedDistanceField::GetFullGradient)
        .def("GetResolution", &SignedDistanceField::GetResolution)
        .def("GetGradient",
             [](const SignedDistanceField& s, int64_t x, int64_t y, int64_t z, bool e) { return s.GetGradient(x, y, z, e); },
             "get the gradient based on index", py::arg("x_index"), py::arg("y_index"), py::arg("z_index"),
             py::arg("enable_edge_gradients") = false)
        .def("GetMessageRepresentation", [](const SignedDistanceField& s) { return SignedDistanceField::GetMessageRepresentation(s); })
        .def_static("LoadFromMessageRepresentation", &SignedDistanceField::LoadFromMessageRepresentation)
        .def("SaveToFile", [](const SignedDistanceField& s, const std::string& path, bool compress) { SignedDistanceField::SaveToFile(s, path, compress); })
        .def_static("LoadFromFile", &SignedDistanceField::LoadFromFile)
        .def("SerializeSelf", [](const SignedDistanceField& s) { std::vector<uint8_t> b; s.SerializeSelf(b); return py::bytes(reinterpret_cast<const char*>(b.data()), b.size()); })
        .def("DeserializeSelf",
             [](SignedDistanceField& s, const std::vector<uint8_t>& buffer, uint64_t current, py::object) { return s.DeserializeSelf(buffer, current); },
             "deserialize", py::arg("buffer"), py::arg("current"), py::arg("value_deserializer") = py::none())
        .def("GetOriginTransform", &SignedDistanceField::GetOriginTransform)
        .def("GetValueByCoordinates", [](const SignedDistanceField& s, double x, double y, double z) { const auto q = s.GetImmutable(x, y, z); return std::make_pair(q.first, q.second); },
             "Please don't mutate this", py::arg("x"), py::arg("y"), py::arg("z"))
        .def("GetValueByIndex", [](const SignedDistanceField& s, int64_t x, int64_t y, int64_t z) { const auto q = s.GetImmutable(x, y, z); return std::make_pair(q.first, q.second); },
             "Please don't mutate this", py::arg("x_index"), py::arg("y_index"), py::arg("z_index"))
        .def("GetNumXCells", &SignedDistanceField::GetNumXCells)
        .def("GetNumYCells", &SignedDistanceField::GetNumYCells)
        .def("GetNumZCells", &SignedDistanceField::GetNumZCells)
        .def("GetFrame", &SignedDistanceField::GetFrame)
        .def("EstimateDistance", [](const SignedDistanceField& s, double x, double y, double z) { return s.EstimateDistance(x, y, z); })
        // fast paths beside the reference-shaped API
        .def("GetRawDataNumpy", [](const SignedDistanceField& s) {
            py::array_t<float> out({s.GetNumXCells(), s.GetNumYCells(), s.GetNumZCells()});
            std::memcpy(out.mutable_data(), s.GetImmutableRawData().data(), s.GetImmutableRawData().size() * sizeof(float));
            return out;
        })
        .def("GetFullGradientNumpy", [](const SignedDistanceField& s, bool enable_edge_gradients) {
            const int64_t nx = s.GetNumXCells(), ny = s.GetNumYCells(), nz = s.GetNumZCells();
            // one kernel on the GPU (sdfgpu_gradient) instead of nx*ny*nz GetGradient calls; same values bit for bit.
            // Cells without a gradient (boundary shell when edge gradients are off) hold 3 x the OOB value here; the
            // reference's GetFullGradient stores an empty vector there (sdf.hpp:341-358).
            std::vector<double> g;
            {
                py::gil_scoped_release release;
                g = s.GetFullGradientFlat(enable_edge_gradients, (double)s.GetOOBValue());
            }
            py::array_t<double> out({nx, ny, nz, (int64_t)3});
            std::memcpy(out.mutable_data(), g.data(), g.size() * sizeof(double));
            return out;
        }, py::arg("enable_edge_gradients") = true)
        .def("GetFullGradientNumpyHost", [](const SignedDistanceField& s, bool enable_edge_gradients) {
            // the reference's per-voxel loop (sdf.hpp:341-358) on one host core: kept as the checker of the GPU path
            const int64_t nx = s.GetNumXCells(), ny = s.GetNumYCells(), nz = s.GetNumZCells();
            py::array_t<double> out({nx, ny, nz, (int64_t)3});
            double* p = out.mutable_data();
            const double oob = (double)s.GetOOBValue();
            for (int64_t x = 0; x < nx; x++) for (int64_t y = 0; y < ny; y++) for (int64_t z = 0; z < nz; z++) {
                const std::vector<double> g = s.GetGradient(x, y, z, enable_edge_gradients);
                for (int k = 0; k < 3; k++) *p++ = g.size() == 3 ? g[(size_t)k] : oob;
            }
            return out;
        }, py::arg("enable_edge_gradients") = true);

    // the field left in HBM (include/sdf_tools/device_sdf.hpp): batched queries without the download.  A pybind thread is the
    // thread that owns the libsdfgpu context, so build, query and drop the object from the same Python thread.
    py::class_<sdf_tools::DeviceSignedDistanceField>(m, "DeviceSignedDistanceField")
        .def("GetResolution", &sdf_tools::DeviceSignedDistanceField::GetResolution)
        .def("GetFrame", &sdf_tools::DeviceSignedDistanceField::GetFrame)
        .def("GetNumXCells", &sdf_tools::DeviceSignedDistanceField::GetNumXCells)
        .def("GetNumYCells", &sdf_tools::DeviceSignedDistanceField::GetNumYCells)
        .def("GetNumZCells", &sdf_tools::DeviceSignedDistanceField::GetNumZCells)
        .def("GetExtrema", &sdf_tools::DeviceSignedDistanceField::GetExtrema)
        .def("HostCopyExists", &sdf_tools::DeviceSignedDistanceField::HostCopyExists)
        .def("DevicePointer", [](sdf_tools::DeviceSignedDistanceField& d) { return (uintptr_t)d.DevicePointer(); },
             "address of the [x][y][z] fp32 field in HBM (e.g. for torch / the *_device ABI)")
        .def("Host", [](const sdf_tools::DeviceSignedDistanceField& d) { return d.Host(); }, "the reference's container (downloads once)")
        .def("QueryBatch", [](const sdf_tools::DeviceSignedDistanceField& d,
                              const py::array_t<double, py::array::c_style | py::array::forcecast>& points, bool enable_edge_gradients) {
            if (points.ndim() != 2 || points.shape(1) != 3) throw std::invalid_argument("points must be [n, 3] float64 (world frame)");
            const int64_t n = points.shape(0);
            py::array_t<double> dist({n}), grad({n, (int64_t)3});
            py::array_t<uint8_t> flags({n});
            {
                py::gil_scoped_release release;
                d.QueryBatch(points.data(), n, enable_edge_gradients, dist.mutable_data(), grad.mutable_data(), flags.mutable_data());
            }
            return py::make_tuple(dist, grad, flags);
        }, py::arg("points"), py::arg("enable_edge_gradients") = false,
           "n x EstimateDistance3d + GetGradient3d (sdf.hpp:947-953, :395-403) in one kernel: (distance [n], gradient [n, 3], flags [n]: "
           "bit 0 inside the grid, bit 1 gradient available)");

    py::class_<CollisionMapGrid>(m, "CollisionMapGrid")
        .def(py::init<Isometry3d const&, std::string, double, int64_t, int64_t, int64_t, COLLISION_CELL const&>())
        .def("SetValue", [](CollisionMapGrid& g, int64_t x, int64_t y, int64_t z, const COLLISION_CELL& c) { return g.SetValue(x, y, z, c); })
        .def("SetValueByCoordinates", [](CollisionMapGrid& g, double x, double y, double z, const COLLISION_CELL& c) { return g.SetValue(x, y, z, c); })
        // wire formats (collision_map.cpp:21-62, :205-315)
        .def(py::init<>())
        .def("SerializeSelf", [](const CollisionMapGrid& g) { std::vector<uint8_t> b; g.SerializeSelf(b); return py::bytes(reinterpret_cast<const char*>(b.data()), b.size()); })
        .def_static("Deserialize", [](const py::bytes& data) { const std::string s = data; CollisionMapGrid g; g.DeserializeSelf(std::vector<uint8_t>(s.begin(), s.end()), 0); return g; })
        .def("SaveToFile", [](const CollisionMapGrid& g, const std::string& path, bool compress) { CollisionMapGrid::SaveToFile(g, path, compress); })
        .def_static("LoadFromFile", &CollisionMapGrid::LoadFromFile)
        .def("GetMessageRepresentation", [](const CollisionMapGrid& g) { return CollisionMapGrid::GetMessageRepresentation(g); })
        .def_static("LoadFromMessageRepresentation", &CollisionMapGrid::LoadFromMessageRepresentation)
        .def("GetFrame", &CollisionMapGrid::GetFrame)
        .def("GetResolution", &CollisionMapGrid::GetResolution)
        .def("GetRawData", &CollisionMapGrid::GetImmutableRawData, "Please don't mutate this")
        .def("GetValueByCoordinates", [](const CollisionMapGrid& g, double x, double y, double z) { const auto q = g.GetImmutable(x, y, z); return std::make_pair(q.first, q.second); },
             "Please don't mutate this", py::arg("x"), py::arg("y"), py::arg("z"))
        .def("GetValueByIndex", [](const CollisionMapGrid& g, int64_t x, int64_t y, int64_t z) { const auto q = g.GetImmutable(x, y, z); return std::make_pair(q.first, q.second); },
             "Please don't mutate this", py::arg("x_index"), py::arg("y_index"), py::arg("z_index"))
        .def("GetNumXCells", &CollisionMapGrid::GetNumXCells)
        .def("GetNumYCells", &CollisionMapGrid::GetNumYCells)
        .def("GetNumZCells", &CollisionMapGrid::GetNumZCells)
        .def("ExtractSignedDistanceField", &CollisionMapGrid::ExtractSignedDistanceField, py::call_guard<py::gil_scoped_release>())
        .def("ExtractSignedDistanceFieldDevice", [](const CollisionMapGrid& g, float oob_value, bool unknown_is_filled, bool add_virtual_border) {
                 auto r = g.ExtractSignedDistanceFieldDevice(oob_value, unknown_is_filled, add_virtual_border);
                 std::unique_ptr<sdf_tools::DeviceSignedDistanceField> field(new sdf_tools::DeviceSignedDistanceField(std::move(r.first)));
                 return py::make_tuple(py::cast(std::move(field)), r.second);
             }, "ExtractSignedDistanceField with the field left in HBM -> (DeviceSignedDistanceField, (max, min))")
        .def("ExtractSignedDistanceFieldViaPredicate", &CollisionMapGrid::ExtractSignedDistanceFieldViaPredicate,
             py::call_guard<py::gil_scoped_release>())
        // the reference's cell-predicate overload (sdf_generation.hpp:422-441) with a Python predicate on the cell
        .def("ExtractSignedDistanceFieldCellPredicate",
             [](const CollisionMapGrid& g, const std::function<bool(const COLLISION_CELL&)>& is_filled_fn, float oob_value) {
                 return sdf_generation::ExtractSignedDistanceField<COLLISION_CELL>(g, is_filled_fn, oob_value, g.GetFrame());
             }, py::arg("is_filled_fn"), py::arg("oob_value"))
        .def("SetOccupancyFromNumpy", [](CollisionMapGrid& g, const py::array_t<float, py::array::c_style | py::array::forcecast>& occ) {
            if (occ.ndim() != 3 || occ.shape(0) != g.GetNumXCells() || occ.shape(1) != g.GetNumYCells() || occ.shape(2) != g.GetNumZCells())
                throw std::invalid_argument("occupancy array must be [nx, ny, nz]");
            const float* p = occ.data();
            auto& cells = g.GetMutableRawData();
            for (size_t i = 0; i < cells.size(); i++) cells[i] = COLLISION_CELL(p[i]);
        });

    m.def("DecompressBytes", &ZlibHelpers::DecompressBytes);
    m.def("DeserializeFixedSizePODFloat", &arc_utilities::DeserializeFixedSizePOD<float>);
    m.def("DeserializeFixedSizePODd", &arc_utilities::DeserializeVectorOfDoubles);
    m.def("SetDevice", [](int device) { sdf_generation::GpuContext::DeviceIndex() = device; }, "GPU used by ExtractSignedDistanceField");
    m.def("SetNumGpus", [](int n) { sdf_generation::MultiGpuContext::SetNumGpus(n); },
          "n > 1: ExtractSignedDistanceField cuts the grid into x slabs over GPUs 0..n-1 (RCCL exchange); 1 = single GPU");

    py::class_<VoxelGridVecd>(m, "VoxelGrid")
        .def(py::init<>())
        .def("GetRawData", &VoxelGridVecd::GetImmutableRawData, "Please don't mutate this")
        .def("GetNumXCells", &VoxelGridVecd::GetNumXCells)
        .def("GetNumYCells", &VoxelGridVecd::GetNumYCells)
        .def("GetNumZCells", &VoxelGridVecd::GetNumZCells)
        .def("GetValueByCoordinates", [](const VoxelGridVecd& g, double x, double y, double z) { const auto q = g.GetImmutable(x, y, z); return std::make_pair(q.first, q.second); },
             "Please don't mutate this", py::arg("x"), py::arg("y"), py::arg("z"))
        .def("GetValueByIndex", [](const VoxelGridVecd& g, int64_t x, int64_t y, int64_t z) { const auto q = g.GetImmutable(x, y, z); return std::make_pair(q.first, q.second); },
             "Please don't mutate this", py::arg("x_index"), py::arg("y_index"), py::arg("z_index"))
        .def("SerializeSelf", [](const VoxelGridVecd& g) { std::vector<uint8_t> b; g.SerializeSelf(b, arc_utilities::SerializeVectorOfDoubles); return py::bytes(reinterpret_cast<const char*>(b.data()), b.size()); })
        .def("DeserializeSelf", [](VoxelGridVecd& g, const std::vector<uint8_t>& buffer, uint64_t current, py::object) {
            return g.DeserializeSelf(buffer, current, arc_utilities::DeserializeVectorOfDoubles); },
             "deserialize", py::arg("buffer"), py::arg("current"), py::arg("value_deserializer") = py::none());
}
