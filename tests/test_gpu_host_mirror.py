"""The C++ host mirror of the reference's callback interface (rdcfes_amd/host/rdc_host.h): a g++-built
driver runs `attach_assemble_function(assemble_<model>)` + `model.assemble()` / `model.solve()`
through the C-ABI; results are compared with the oracle.  Reads like the reference's own driver
(src/pihna.C:18-96) because the names and call order are the same."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

from rdcfes_amd import (adpm_params_from_dict, hcc_params_from_dict, pihna_params_from_dict, proteas_params_from_dict,
                        ripf_params_from_dict, synth)

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def driver():
    from rdcfes_amd import build
    lib = build.build(verbose=False)
    out = ROOT / "tests" / "_build" / "host_mirror_driver"
    out.parent.mkdir(exist_ok=True)
    src = ROOT / "tests" / "host_mirror_driver.cpp"
    hdr = ROOT / "rdcfes_amd" / "host" / "rdc_host.h"
    if not out.exists() or out.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime, lib.stat().st_mtime):
        subprocess.run(["g++", "-O2", "-std=c++17", str(src), "-o", str(out), f"-L{lib.parent}", "-lrdc_assembly",
                        f"-Wl,-rpath,{lib.parent}"], check=True)
    return out


def _write_case(d, conn, xyz, u, params, extra=None):
    conn.astype(np.uint32).tofile(d / "conn.bin")
    xyz.astype(np.float64).tofile(d / "xyz.bin")
    u.astype(np.float64).tofile(d / "u.bin")
    (d / "params.txt").write_text("".join(f"{k} {v!r}\n" for k, v in params.items()))
    for name, arr in (extra or {}).items():
        arr.astype(np.float64).tofile(d / name)


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


@pytest.mark.parametrize("model,nen", [("pihna", 4), ("ripf", 4), ("hcc", 8), ("adpm", 4), ("proteas", 8)])
def test_callback_through_host_mirror(oracle, driver, tmp_path, model, nen):
    conn, xyz = synth.kuhn_tet_mesh(6, order="random") if nen == 4 else synth.hex_mesh(6, jitter=0.1, order="random")
    extra, aux, tracts = {}, None, None
    if model == "pihna":
        d = synth.pihna_param_dict("full")
        p, u, mid, nv = pihna_params_from_dict(d), synth.pihna_fields(xyz), 0, 5
    elif model == "ripf":
        d = synth.ripf_param_dict("full")
        d["volume_fraction/max_vacant"] = 0.5
        p, mid, nv = ripf_params_from_dict(d), 1, 3
        u, aux = synth.ripf_fields(xyz)
        td = np.column_stack([np.zeros(len(u)), aux[:, 0], aux[:, 1]])
        rt = np.column_stack([np.zeros(len(u)), np.zeros(len(u)), aux[:, 2]])
        extra = {"td.bin": td, "rt.bin": rt}
        from rdcfes_amd.params import RIPF_DEFAULTS
        d = {**RIPF_DEFAULTS, **d}
    elif model == "adpm":
        import math
        from rdcfes_amd.params import ADPM_DEFAULTS
        d = {**ADPM_DEFAULTS, **synth.adpm_param_dict("full")}
        p, mid, nv = adpm_params_from_dict(synth.adpm_param_dict("full"), time=3.0), 4, 3
        u, tracts = synth.adpm_fields(xyz, conn.shape[0])
        extra = {"tracts.bin": tracts}
        d = dict(d)
        d["taxis/A_b/angle"] = math.radians(d["taxis/A_b/angle"])   # es.parameters holds radians (src/adpm.C:193)
        d["taxis/Tau/angle"] = math.radians(d["taxis/Tau/angle"])
        d["time"] = 3.0
    elif model == "proteas":
        from rdcfes_amd.params import PROTEAS_DEFAULTS
        d = {**PROTEAS_DEFAULTS, **synth.proteas_param_dict("full")}
        p, mid, nv = proteas_params_from_dict(synth.proteas_param_dict("full")), 5, 5
        u, aux = synth.proteas_fields(xyz)
        extra = {"aux2.bin": aux[:, :2]}
    else:
        d = synth.hcc_param_dict("full")
        p, u, mid, nv = hcc_params_from_dict(d), synth.hcc_fields(xyz), 2, 3
    if model == "pihna":
        from rdcfes_amd.params import PIHNA_DEFAULTS
        d = {**PIHNA_DEFAULTS, **d}
    if model == "hcc":
        from rdcfes_amd.params import HCC_DEFAULTS
        d = {**HCC_DEFAULTS, **d}
    if model == "pihna":
        d = {**d, "range/active_tumor/min": 100.0, "range/active_tumor/max": 1e9, "range/necrotic/min": 50.0, "range/necrotic/max": 1e9,
             "range/vascularity/min": 0.0, "range/vascularity/max": 7000.0, "range/total_cell/min": 0.031, "range/total_cell/max": 1.0}
    if model == "ripf":
        d = {**d, "range_cc/HU/min": -1e9, "range_cc/HU/max": 1e9, "range_cc/min": float(np.median(u[:, 1])),
             "range_fb/HU/min": -1e9, "range_fb/HU/max": 1e9, "range_fb/min": float(np.median(u[:, 2]))}
    _write_case(tmp_path, conn, xyz, u, d, extra)
    r = subprocess.run([str(driver), str(tmp_path), model, str(nen)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    if model == "pihna":     # save_solution: header + one line (time 0), volumes from the device
        from rdcfes_amd import PihnaRanges
        head, line = (tmp_path / "out.csv").read_text().splitlines()
        assert head.startswith('"TIME","DEGREES_OF_FREEDOM","ACTIVE_TUMOR_VOLUME"')
        f = [float(x) for x in line.split(",")]
        v0 = oracle.pihna_volume_integrals(nen, conn, xyz, u, PihnaRanges(100.0, 1e9, 50.0, 1e9, 0.0, 7000.0, 0.031, 1.0, d["cells_max_capacity"]))
        assert f[0] == 0.0 and f[1] == 5 * xyz.shape[0]
        np.testing.assert_allclose(f[2:], v0, rtol=1e-12, atol=1e-15)
        assert 0.0 < v0[0] < 1.0
    if model == "ripf":
        from rdcfes_amd import RipfRanges
        (line,) = (tmp_path / "out.csv").read_text().splitlines()
        f = [float(x) for x in line.split(",")]
        v0 = oracle.ripf_volume_integrals(nen, conn, xyz, u, RipfRanges(-1e9, 1e9, d["range_cc/min"], -1e9, 1e9, d["range_fb/min"]))
        np.testing.assert_allclose(f[1:], v0, rtol=1e-12, atol=1e-15)
        assert np.all(v0 > 0.0) and np.all(v0 < 1.0)
    val = np.fromfile(tmp_path / "val.bin")
    rhs = np.fromfile(tmp_path / "rhs.bin")
    rp = np.fromfile(tmp_path / "row_ptr.bin", dtype=np.int64)
    col = np.fromfile(tmp_path / "col_idx.bin", dtype=np.int32)
    rp0, col0, val0, rhs0 = oracle.assemble(mid, nen, conn, xyz, nv, p, u_old=u, aux=aux, elem_fibre=tracts)
    np.testing.assert_array_equal(rp, rp0)
    np.testing.assert_array_equal(col, col0)
    assert rel(val, val0) < 1e-10 and rel(rhs, rhs0) < 1e-10


def test_time_step_through_host_mirror(oracle, driver, tmp_path):
    """one implicit step: model.solve() = assemble (GPU) + linear solve (host stand-in for PETSc KSP)"""
    conn, xyz = synth.kuhn_tet_mesh(4, order="lex")
    from rdcfes_amd.params import PIHNA_DEFAULTS
    d = {**PIHNA_DEFAULTS, **synth.pihna_param_dict("shipped")}
    u = synth.pihna_fields(xyz)
    _write_case(tmp_path, conn, xyz, u, d)
    r = subprocess.run([str(driver), str(tmp_path), "pihna", "4", "solve"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "relative residual" in r.stdout
    res = float(r.stdout.split("relative residual")[1])
    assert res < 1e-9
    sol = np.fromfile(tmp_path / "solution.bin")
    # oracle system, solved with scipy, must give the same new state
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    p = pihna_params_from_dict(synth.pihna_param_dict("shipped"))
    rp0, col0, val0, rhs0 = oracle.assemble(0, 4, conn, xyz, 5, p, u_old=u)
    A = sp.csr_matrix((val0, col0, rp0), shape=(rhs0.size, rhs0.size))
    x = spla.spsolve(A.tocsc(), rhs0)
    assert np.linalg.norm(sol - x) / np.linalg.norm(x) < 1e-8


@pytest.mark.parametrize("chunks", [2, 7])
def test_chunked_handback_through_host_mirror(oracle, driver, tmp_path, chunks):
    """"rdc/handback_chunks" > 1: the pipelined hand-back of the libMesh adapter (rdc_csr_download_rows_async on the context's
    copy stream, two chunks in flight, rdc_ticket_wait, one consumer call per chunk) mirrored in rdc_host.h: every node range
    is delivered once, in order, into the same positions of the matrix / rhs arrays as the whole download."""
    conn, xyz = synth.kuhn_tet_mesh(7, order="random")
    from rdcfes_amd.params import PIHNA_DEFAULTS
    d = {**PIHNA_DEFAULTS, **synth.pihna_param_dict("shipped"), "rdc/handback_chunks": chunks}
    u = synth.pihna_fields(xyz)
    _write_case(tmp_path, conn, xyz, u, d)
    r = subprocess.run([str(driver), str(tmp_path), "pihna", "4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    log = np.fromfile(tmp_path / "handback_log.bin", dtype=np.int64).reshape(-1, 2)
    nn = xyz.shape[0]
    assert log.shape[0] == chunks and log[0, 0] == 0 and log[-1, 1] == nn
    assert np.array_equal(log[1:, 0], log[:-1, 1]) and np.all(log[:, 1] >= log[:, 0])      # a partition of [0, n_nodes), in order
    val, rhs = np.fromfile(tmp_path / "val.bin"), np.fromfile(tmp_path / "rhs.bin")
    p = pihna_params_from_dict(synth.pihna_param_dict("shipped"))
    _, _, val0, rhs0 = oracle.assemble(0, 4, conn, xyz, 5, p, u_old=u)
    assert rel(val, val0) < 1e-10 and rel(rhs, rhs0) < 1e-10
