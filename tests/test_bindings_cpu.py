"""CPU: the pysdf_tools surface (names / argument orders of the reference's bindings.cpp:15-106)
and the host-side containers.  No SDF is computed here (that needs the GPU)."""
import numpy as np
import pytest

from sdf_tools_amd import capi
from sdf_tools_amd._bindings import load_pysdf_tools

m = load_pysdf_tools()
IDENT = [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]]


def test_module_surface_matches_reference_bindings():
    for cls, methods in {
        "COLLISION_CELL": ["occupancy", "component"],
        "Isometry3d": ["translation"],
        "SignedDistanceField": ["GetRawData", "GetFullGradient", "GetResolution", "GetGradient",
                                "GetMessageRepresentation", "LoadFromMessageRepresentation", "SaveToFile",
                                "LoadFromFile", "SerializeSelf", "DeserializeSelf", "GetOriginTransform",
                                "GetValueByCoordinates", "GetValueByIndex", "GetNumXCells", "GetNumYCells", "GetNumZCells"],
        "CollisionMapGrid": ["SetValue", "SetValueByCoordinates", "GetRawData", "GetValueByCoordinates",
                             "GetValueByIndex", "GetNumXCells", "GetNumYCells", "GetNumZCells",
                             "ExtractSignedDistanceField", "SerializeSelf", "SaveToFile", "LoadFromFile",
                             "GetMessageRepresentation", "LoadFromMessageRepresentation"],
        "CollisionMap": ["serialized_map", "is_compressed", "frame_id"],
        "VoxelGrid": ["GetRawData", "GetNumXCells", "GetNumYCells", "GetNumZCells", "GetValueByCoordinates",
                      "GetValueByIndex", "SerializeSelf", "DeserializeSelf"],
    }.items():
        c = getattr(m, cls)
        for name in methods:
            assert hasattr(c, name), (cls, name)
    for fn in ("DecompressBytes", "DeserializeFixedSizePODFloat", "DeserializeFixedSizePODd"):
        assert hasattr(m, fn)


def test_collision_cell_and_isometry():
    c = m.COLLISION_CELL(0.75)
    assert c.occupancy == 0.75 and c.component == 0
    c = m.COLLISION_CELL(1.0, 7)
    assert c.component == 7
    t = m.Isometry3d([[1, 0, 0, -5.0], [0, 1, 0, 2.5], [0, 0, 1, 1.0], [0, 0, 0, 1]])
    assert t.translation().tolist() == [-5.0, 2.5, 1.0]


def test_collision_map_grid_set_get_and_location_convention():
    origin = m.Isometry3d([[1, 0, 0, -1.0], [0, 1, 0, -2.0], [0, 0, 1, 0.0], [0, 0, 0, 1]])
    g = m.CollisionMapGrid(origin, "world", 0.5, 4, 6, 2, m.COLLISION_CELL(-10000))
    assert (g.GetNumXCells(), g.GetNumYCells(), g.GetNumZCells()) == (4, 6, 2)
    cell, ok = g.GetValueByIndex(0, 0, 0)
    assert ok and cell.occupancy == -10000
    assert g.SetValue(3, 5, 1, m.COLLISION_CELL(1.0)) is True
    assert g.SetValue(4, 0, 0, m.COLLISION_CELL(1.0)) is False          # out of bounds
    cell, ok = g.GetValueByIndex(3, 5, 1)
    assert ok and cell.occupancy == 1.0
    cell, ok = g.GetValueByIndex(-1, 0, 0)
    assert not ok and cell.occupancy == -10000                         # OOB value
    # cell index = floor((location - origin) / resolution); centre at (i + 0.5) * resolution
    assert g.SetValueByCoordinates(-1.0 + 0.5 * 2 + 0.01, -2.0 + 0.5 * 3 + 0.4, 0.6, m.COLLISION_CELL(0.9))
    cell, ok = g.GetValueByIndex(2, 3, 1)
    assert ok and abs(cell.occupancy - 0.9) < 1e-6
    cell, ok = g.GetValueByCoordinates(-1.0 + 1.25, -2.0 + 1.75, 0.75)
    assert ok and abs(cell.occupancy - 0.9) < 1e-6
    raw = g.GetRawData()
    assert len(raw) == 48 and raw[2 * 12 + 3 * 2 + 1].occupancy == pytest.approx(0.9)   # x*ny*nz + y*nz + z


def test_set_occupancy_from_numpy():
    g = m.CollisionMapGrid(m.Isometry3d(IDENT), "w", 1.0, 3, 4, 5, m.COLLISION_CELL(0.0))
    occ = np.arange(60, dtype=np.float32).reshape(3, 4, 5)
    g.SetOccupancyFromNumpy(occ)
    assert g.GetValueByIndex(2, 1, 3)[0].occupancy == occ[2, 1, 3]
    with pytest.raises(ValueError):
        g.SetOccupancyFromNumpy(np.zeros((4, 3, 5), np.float32))


def test_extract_without_gpu_raises_not_falls_back():
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    g = m.CollisionMapGrid(m.Isometry3d(IDENT), "w", 1.0, 4, 4, 4, m.COLLISION_CELL(0.0))
    with pytest.raises(RuntimeError) as ei:
        g.ExtractSignedDistanceField(-10000.0, False, False)
    assert "no CPU fallback" in str(ei.value)


def test_zlib_and_pod_helpers():
    import struct
    import zlib
    payload = bytes(range(200)) * 10
    assert bytes(m.DecompressBytes(list(zlib.compress(payload)))) == payload
    val, used = m.DeserializeFixedSizePODFloat(list(struct.pack("<f", 1.5)), 0)
    assert val == 1.5 and used == 4
    vec, used = m.DeserializeFixedSizePODd(list(struct.pack("<Qddd", 3, 1.0, 2.0, 3.0)), 0)
    assert vec == [1.0, 2.0, 3.0] and used == 32


def test_empty_sdf_serialisation_round_trip():
    s = m.SignedDistanceField()
    blob = s.SerializeSelf()
    s2 = m.SignedDistanceField()
    assert s2.DeserializeSelf(list(blob), 0) == len(blob)


def _golden_grid():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "make_collision_map_golden", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_collision_map_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.make_grid(m)


def _cells(g):
    return [(g.GetValueByIndex(x, y, z)[0].occupancy, g.GetValueByIndex(x, y, z)[0].component)
            for x in range(g.GetNumXCells()) for y in range(g.GetNumYCells()) for z in range(g.GetNumZCells())]


def test_collision_map_wire_format_is_pinned_and_round_trips(tmp_path):
    """N3, collision-map half (reference src/sdf_tools/collision_map.cpp:21-62, :205-315): the serialised bytes of a fixed grid
    equal the committed self-golden vector (field order: initialized, both transforms, cell vector, cell / grid sizes,
    strides and counts, default and OOB cell, number of components, frame, components_valid), and every wire form --
    raw bytes, CMGR / CMGZ files, the compressed message -- restores the same grid.  Interoperability with files written
    by the reference itself is unverified (arc_utilities is not vendored there)."""
    import os
    import struct
    g = _golden_grid()
    blob = bytes(g.SerializeSelf())
    golden = bytes.fromhex(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "collision_map_serialized.hex")).read().strip())
    assert blob == golden
    # spot-check the layout: 1 byte initialized, 2 x 16 doubles of transforms, then the u64 cell count and 8-byte cells
    assert blob[0] == 1 and struct.unpack_from("<Q", blob, 1 + 256)[0] == 12
    occ, comp = struct.unpack_from("<fI", blob, 1 + 256 + 8 + 8 * 5)
    assert (occ, comp) == (0.125 * 5, 15)
    assert blob.endswith(struct.pack("<I", 0) + struct.pack("<Q", 12) + b"golden_frame" + b"\x00")
    want = _cells(g)
    back = m.CollisionMapGrid.Deserialize(blob)
    assert _cells(back) == want and back.GetFrame() == "golden_frame" and back.GetResolution() == 0.5
    cell, ok = back.GetValueByIndex(-1, 0, 0)
    assert not ok and (cell.occupancy, cell.component) == (-7.5, 9)
    for compress, magic in ((False, b"CMGR"), (True, b"CMGZ")):
        path = str(tmp_path / ("map_%d.cmg" % compress))
        g.SaveToFile(path, compress)
        raw = open(path, "rb").read()
        assert raw[:4] == magic and (compress or raw[4:] == blob)
        assert _cells(m.CollisionMapGrid.LoadFromFile(path)) == want
    with pytest.raises(Exception):
        m.CollisionMapGrid.LoadFromFile(str(tmp_path / "missing.cmg"))
    bad = str(tmp_path / "bad.cmg")
    open(bad, "wb").write(b"XXXX" + blob)
    with pytest.raises(Exception):
        m.CollisionMapGrid.LoadFromFile(bad)
    msg = g.GetMessageRepresentation()
    assert msg.is_compressed and msg.frame_id == "golden_frame"
    assert bytes(m.DecompressBytes(list(msg.serialized_map))) == blob
    assert _cells(m.CollisionMapGrid.LoadFromMessageRepresentation(msg)) == want
    plain = m.CollisionMap()
    plain.serialized_map = list(blob)
    plain.is_compressed = False
    assert _cells(m.CollisionMapGrid.LoadFromMessageRepresentation(plain)) == want
