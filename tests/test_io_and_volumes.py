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


# ---- RIPF and ADPM CSV reductions (src/ripf.C:777-866, src/adpm.C:690-829) ---------------------------------------
def _ripf_state(xyz):
    rng = np.random.default_rng(11)
    r = np.linalg.norm(xyz - 0.5, axis=1)
    u = np.empty((xyz.shape[0], 3))
    u[:, 0] = -900.0 + 1200.0 * np.exp(-6.0 * r * r) + 20.0 * rng.standard_normal(xyz.shape[0])   # HU
    u[:, 1] = 0.8 * np.exp(-10.0 * r * r)                                                           # cc
    u[:, 2] = 0.5 * np.exp(-4.0 * (r - 0.3) ** 2)                                                   # fb
    return u


def _ripf_ranges():
    from rdcfes_amd import RipfRanges
    return RipfRanges(-500.0, 400.0, 0.2, -800.0, 300.0, 0.25)


def _adpm_case(nen, n):
    conn, xyz = synth.kuhn_tet_mesh(n, jitter=0.1, order="random") if nen == 4 else synth.hex_mesh(n, jitter=0.1, order="random")
    rng = np.random.default_rng(5)
    u = np.abs(rng.standard_normal((xyz.shape[0], 3))) * np.array([1.0, 0.4, 0.2]) + 0.3 * xyz
    cen = xyz[conn.astype(np.int64)].mean(axis=1)
    sub = (1001 + 7 * (np.floor(cen[:, 0] * 3).astype(np.int32) + 3 * np.floor(cen[:, 1] * 2).astype(np.int32))).astype(np.int32)
    ids = np.unique(np.concatenate([sub, [5000]])).astype(np.int32)      # one listed region has no element
    return conn, xyz, u, sub, ids


@pytest.mark.parametrize("nen", [4, 8])
def test_oracle_ripf_and_adpm_reductions(oracle, nen):
    from rdcfes_amd import AdpmRanges, RipfRanges
    conn, xyz = synth.kuhn_tet_mesh(5, jitter=0.1) if nen == 4 else synth.hex_mesh(5, jitter=0.1)
    u = _ripf_state(xyz)
    every = RipfRanges(-1e300, 1e300, -1e300, -1e300, 1e300, -1e300)
    np.testing.assert_allclose(oracle.ripf_volume_integrals(nen, conn, xyz, u, every), 1.0, rtol=1e-12)
    assert np.all(oracle.ripf_volume_integrals(nen, conn, xyz, u, RipfRanges(1.0, -1.0, 0.0, 1.0, -1.0, 0.0)) == 0.0)
    v = oracle.ripf_volume_integrals(nen, conn, xyz, u, _ripf_ranges())
    assert np.all(v > 0.0) and np.all(v < 1.0)
    # ADPM: a linear field is reproduced by the element average; region volumes add up to the cube; the last element wins
    conn, xyz, u, sub, ids = _adpm_case(nen, 4)
    u[:, 1] = 2.0 + xyz[:, 0] - 3.0 * xyz[:, 2]
    out = oracle.adpm_parcellation_integrals(nen, conn, xyz, u, AdpmRanges(-1e300, 1e300, -1e300, 1e300), sub, ids)
    np.testing.assert_allclose(out[:, 2].sum(), 1.0, rtol=1e-12)
    np.testing.assert_allclose(out[:, 2], out[:, 3], rtol=0, atol=0)
    assert np.all(out[ids == 5000] == 0.0)
    for i, ID in enumerate(ids[:-1]):
        e = np.flatnonzero(sub == ID)[-1]
        if nen == 4:   # volume-weighted centroid of a tet = mean of its nodes: exact for the linear field
            cen = xyz[conn[e].astype(np.int64)].mean(axis=0)
            np.testing.assert_allclose(out[i, 0], 2.0 + cen[0] - 3.0 * cen[2], rtol=1e-12)
        lo, hi = u[conn[e].astype(np.int64), 1].min(), u[conn[e].astype(np.int64), 1].max()
        assert lo <= out[i, 0] <= hi


@pytest.mark.gpu
@pytest.mark.parametrize("nen", [4, 8])
def test_gpu_ripf_volume_integrals(oracle, nen):
    from rdcfes_amd import AssemblyContext, FIELD_OLD_SOLUTION
    conn, xyz = synth.kuhn_tet_mesh(9, jitter=0.1, order="random") if nen == 4 else synth.hex_mesh(8, jitter=0.1, order="random")
    u = _ripf_state(xyz)
    half = conn.shape[0] // 2
    v0 = oracle.ripf_volume_integrals(nen, conn, xyz, u, _ripf_ranges())
    v0h = oracle.ripf_volume_integrals(nen, conn, xyz, u, _ripf_ranges(), n_elem=half)
    assert np.all(v0 > 0.0)
    with AssemblyContext(0) as ctx:
        ctx.mesh_upload(nen, conn, xyz, 3)
        ctx.field_upload(FIELD_OLD_SOLUTION, u)
        v = ctx.ripf_volume_integrals(_ripf_ranges())
        vh = ctx.ripf_volume_integrals(_ripf_ranges(), half)
    np.testing.assert_allclose(v, v0, rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(vh, v0h, rtol=1e-12, atol=1e-15)


@pytest.mark.gpu
@pytest.mark.parametrize("nen", [4, 8])
def test_gpu_adpm_parcellation_integrals(oracle, nen):
    from rdcfes_amd import AdpmRanges, AssemblyContext, FIELD_OLD_SOLUTION, RdcError
    conn, xyz, u, sub, ids = _adpm_case(nen, 9 if nen == 4 else 8)
    rg = AdpmRanges(0.1, 0.9, 0.05, 0.45)
    half = conn.shape[0] // 2
    o0 = oracle.adpm_parcellation_integrals(nen, conn, xyz, u, rg, sub, ids)
    o0h = oracle.adpm_parcellation_integrals(nen, conn, xyz, u, rg, sub, ids, n_elem=half)
    assert (o0[:, 2] > 0).sum() >= 3 and (o0[:, 3] > 0).sum() >= 3
    with AssemblyContext(0) as ctx:
        ctx.mesh_upload(nen, conn, xyz, 3)
        ctx.field_upload(FIELD_OLD_SOLUTION, u)
        o, last = ctx.adpm_parcellation_integrals(rg, sub, ids)
        oh, lasth = ctx.adpm_parcellation_integrals(rg, sub, ids, half)
        with pytest.raises(RdcError):
            ctx.adpm_parcellation_integrals(rg, sub, ids[::-1])          # the parcellation is an ordered set
    np.testing.assert_allclose(o, o0, rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(oh, o0h, rtol=1e-12, atol=1e-15)
    for i, ID in enumerate(ids):
        w = np.flatnonzero(sub == ID)
        assert last[i] == (w[-1] if w.size else -1)
        wh = w[w < half]
        assert lasth[i] == (wh[-1] if wh.size else -1)
