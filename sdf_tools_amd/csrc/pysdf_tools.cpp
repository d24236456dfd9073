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
