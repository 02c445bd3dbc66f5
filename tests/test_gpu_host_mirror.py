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
    _write_case(tmp_path, conn, xyz, u, d, extra)
    r = subprocess.run([str(driver), str(tmp_path), model, str(nen)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
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
