"""nodal/elemental .dat files, the VTU/PVD writer, and the CSV volume integrals."""
import xml.etree.ElementTree as ET

import numpy as np
import pytest

from rdcfes_amd import PihnaRanges, io, synth


def test_field_dat_roundtrip(tmp_path):
    a = np.random.default_rng(0).standard_normal((37, 5))
    io.write_field_dat(tmp_path / "nodal.dat", a)
    np.testing.assert_array_equal(io.read_field_dat(tmp_path / "nodal.dat", 37, 5), a)
    (tmp_path / "free.dat").write_text("1 2\n3\n4 5 6\n")          # `fin >>` does not care about line breaks
    np.testing.assert_array_equal(io.read_field_dat(tmp_path / "free.dat", 3, 2), [[1, 2], [3, 4], [5, 6]])
    with pytest.raises(ValueError):
        io.read_field_dat(tmp_path / "free.dat", 4, 2)


@pytest.mark.parametrize("nen", [4, 8])
def test_vtu_and_pvd(tmp_path, nen):
    conn, xyz = synth.kuhn_tet_mesh(2) if nen == 4 else synth.hex_mesh(2)
    xyz = np.vstack([xyz, [[9.0, 9.0, 9.0]]])                      # an orphan node: left out, as upstream
    names = ["n", "c", "h", "v", "a"]
    u = np.random.default_rng(1).uniform(0, 1, (xyz.shape[0], 5))
    u[0, 0] = 1e-40                                               # below SMALLEST_NUMBER -> 0
    pvd = io.PvdCollection(tmp_path / "out")
    f0 = pvd.add(0.0, nen, conn, xyz, names, u, region_id=np.arange(conn.shape[0]) % 3)
    pvd.add(0.5, nen, conn, xyz, names, u)
    pvd.close()
    root = ET.parse(f0).getroot()
    piece = root.find("UnstructuredGrid/Piece")
    assert int(piece.get("NumberOfPoints")) == xyz.shape[0] - 1 and int(piece.get("NumberOfCells")) == conn.shape[0]
    arrays = {a.get("Name"): np.array(a.text.split(), dtype=float) for a in root.iter("DataArray")}
    assert list(arrays) == ["position", "node_ID"] + names + ["element_ID", "region_ID", "processor_ID", "connectivity", "offsets", "types"]
    np.testing.assert_allclose(arrays["position"].reshape(-1, 3), xyz[:-1])
    np.testing.assert_array_equal(arrays["connectivity"].reshape(-1, nen), conn)
    np.testing.assert_array_equal(arrays["offsets"], nen * (np.arange(conn.shape[0]) + 1))
    assert set(arrays["types"]) == {10.0 if nen == 4 else 12.0} and arrays["n"][0] == 0.0
    np.testing.assert_allclose(arrays["c"], u[:-1, 1])
    coll = ET.parse(tmp_path / "out.pvd").getroot()
    assert [d.get("file") for d in coll.iter("DataSet")] == ["out_000000.vtu", "out_000001.vtu"]


def _ranges():
    return PihnaRanges(100.0, 1.0e9, 50.0, 1.0e9, 0.0, 7000.0, 0.031, 1.0, 2.39e5)


@pytest.mark.parametrize("nen", [4, 8])
def test_oracle_volume_integrals(oracle, nen):
    conn, xyz = synth.kuhn_tet_mesh(5, jitter=0.1) if nen == 4 else synth.hex_mesh(5, jitter=0.1)
    u = synth.pihna_fields(xyz)
    every = PihnaRanges(-1e300, 1e300, -1e300, 1e300, -1e300, 1e300, -1e300, 1e300, 2.39e5)
    np.testing.assert_allclose(oracle.pihna_volume_integrals(nen, conn, xyz, u, every), 1.0, rtol=1e-12)   # unit cube
    none = PihnaRanges(1.0, -1.0, 1.0, -1.0, 1.0, -1.0, 1.0, -1.0, 2.39e5)
    assert np.all(oracle.pihna_volume_integrals(nen, conn, xyz, u, none) == 0.0)
    v = oracle.pihna_volume_integrals(nen, conn, xyz, u, _ranges())
    assert np.all(v >= 0.0) and np.all(v <= 1.0) and 0.0 < v[0] < 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("nen", [4, 8])
def test_gpu_volume_integrals(oracle, nen):
    from rdcfes_amd import AssemblyContext, FIELD_OLD_SOLUTION
    conn, xyz = synth.kuhn_tet_mesh(9, jitter=0.1, order="random") if nen == 4 else synth.hex_mesh(8, jitter=0.1, order="random")
    u = synth.pihna_fields(xyz)
    v0 = oracle.pihna_volume_integrals(nen, conn, xyz, u, _ranges())
    half = conn.shape[0] // 2
    v0h = oracle.pihna_volume_integrals(nen, conn, xyz, u, _ranges(), n_elem=half)
    with AssemblyContext(0) as ctx:
        ctx.mesh_upload(nen, conn, xyz, 5)
        ctx.field_upload(FIELD_OLD_SOLUTION, u)
        v = ctx.pihna_volume_integrals(_ranges())
        vh = ctx.pihna_volume_integrals(_ranges(), half)
    np.testing.assert_allclose(v, v0, rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(vh, v0h, rtol=1e-12, atol=1e-15)
