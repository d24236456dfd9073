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


def test_tagged_map_wire_format_is_pinned_and_round_trips(tmp_path):
    """N3 / N4 symmetry (VERDICT r3 "next round" 8; reference src/sdf_tools/tagged_object_collision_map.cpp:23-75, :242-339,
    msg/TaggedObjectCollisionMap.msg): the serialised bytes of a fixed tagged map equal the committed self-golden (field
    order: the VoxelGrid block with 16-byte cells, then number of components, number of convex segments, frame,
    components_valid, convex_segments_valid), and raw bytes, TCMR / TCMZ files and the compressed message restore the same
    map.  Interoperability with files written by the reference itself is unverified (arc_utilities is not vendored)."""
    import importlib.util
    import os
    import struct
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_tagged_map_golden", os.path.join(here, "golden", "make_tagged_map_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = mod.make_grid(m)
    blob = bytes(g.SerializeSelf())
    golden = bytes.fromhex(open(os.path.join(here, "golden", "tagged_map_serialized.hex")).read().strip())
    assert blob == golden

    def cells(t):
        out = []
        for x in range(t.GetNumXCells()):
            for y in range(t.GetNumYCells()):
                for z in range(t.GetNumZCells()):
                    c = t.GetValueByIndex(x, y, z)[0]
                    out.append((c.occupancy, c.component, c.object_id, c.convex_segment))
        return out

    # layout: 1 byte initialized, 2 x 16 doubles of transforms, u64 cell count, 16-byte cells {occupancy, component, object, segment}
    assert blob[0] == 1 and struct.unpack_from("<Q", blob, 1 + 256)[0] == 12
    assert struct.unpack_from("<fIII", blob, 1 + 256 + 8 + 16 * 5) == (0.25 * 5, 105, 1, 35)
    assert blob.endswith(struct.pack("<II", 0, 0) + struct.pack("<Q", 13) + b"tagged_golden" + b"\x00\x00")
    want = cells(g)
    back = m.TaggedObjectCollisionMapGrid.Deserialize(blob)
    assert cells(back) == want and back.GetFrame() == "tagged_golden" and back.GetResolution() == 0.25
    oob, ok = back.GetValueByIndex(0, 3, 0)
    assert not ok and (oob.occupancy, oob.object_id, oob.component, oob.convex_segment) == (-3.5, 11, 5, 2)
    for compress, magic in ((False, b"TCMR"), (True, b"TCMZ")):
        path = str(tmp_path / ("map_%d.tcm" % compress))
        g.SaveToFile(path, compress)
        raw = open(path, "rb").read()
        assert raw[:4] == magic and (compress or raw[4:] == blob)
        assert cells(m.TaggedObjectCollisionMapGrid.LoadFromFile(path)) == want
    bad = str(tmp_path / "bad.tcm")
    open(bad, "wb").write(b"CMGR" + blob)                 # the plain map's magic is not this map's
    with pytest.raises(Exception):
        m.TaggedObjectCollisionMapGrid.LoadFromFile(bad)
    msg = g.GetMessageRepresentation()
    assert msg.is_compressed and msg.frame_id == "tagged_golden"
    assert cells(m.TaggedObjectCollisionMapGrid.LoadFromMessageRepresentation(msg)) == want
    msg2 = m.TaggedObjectCollisionMap()
    msg2.serialized_map, msg2.is_compressed = list(blob), False
    assert cells(m.TaggedObjectCollisionMapGrid.LoadFromMessageRepresentation(msg2)) == want


def test_deserialisers_refuse_hostile_headers():
    """Deserialize / LoadFromFile / LoadFromMessageRepresentation take untrusted bytes.  An element count beyond
    the buffer must be refused before memory is reserved for it; negative or overflowing dimensions must not pass the
    consistency check; truncated buffers raise instead of reading past the end."""
    import struct
    g = _golden_grid()
    blob = bytearray(bytes(g.SerializeSelf()))
    count_at = 1 + 256
    huge = bytearray(blob)
    struct.pack_into("<Q", huge, count_at, 1 << 60)       # "2^60 cells follow"
    with pytest.raises(Exception):
        m.CollisionMapGrid.Deserialize(bytes(huge))
    # dims block: 9 doubles then stride1, stride2, nx, ny, nz (int64) behind the 12 cells
    dims_at = count_at + 8 + 12 * 8 + 9 * 8 + 2 * 8
    assert struct.unpack_from("<qqq", blob, dims_at) == (3, 2, 2)
    for bad_dims in ((-3, -2, 2), (3, 2, -2), (1 << 62, 1 << 62, 48), (0, 0, 0), (12, 1, 1)):
        b = bytearray(blob)
        struct.pack_into("<qqq", b, dims_at, *bad_dims)
        with pytest.raises(Exception):
            m.CollisionMapGrid.Deserialize(bytes(b))
    for cut in (0, 1, 100, count_at + 4, len(blob) - 1):
        with pytest.raises(Exception):
            m.CollisionMapGrid.Deserialize(bytes(blob[:cut]))
    frame_len_at = len(blob) - 1 - 12 - 8
    assert struct.unpack_from("<Q", blob, frame_len_at)[0] == 12
    b = bytearray(blob)
    struct.pack_into("<Q", b, frame_len_at, (1 << 64) - 4)    # string length that wraps the bounds arithmetic
    with pytest.raises(Exception):
        m.CollisionMapGrid.Deserialize(bytes(b))
