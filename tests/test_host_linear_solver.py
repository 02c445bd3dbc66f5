"""The host mirror's stand-in for PETSc KSP (ILU(0)-preconditioned BiCGStab, rdcfes_amd/host/rdc_host.h) on the two
kinds of systems the drivers hand it: the penalty-stiffened, non-symmetric Jacobian of the solid system on the
reference's shipped cube, and a reaction-diffusion matrix.  CPU only (no C-ABI call is made)."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

from rdcfes_amd import gmsh, hcc_params_from_dict, inputs, synth

ROOT = Path(__file__).resolve().parent.parent
G = ROOT / "tests" / "golden"


@pytest.fixture(scope="module")
def solver():
    out = ROOT / "tests" / "_build" / "host_linear_solver"
    out.parent.mkdir(exist_ok=True)
    src = ROOT / "tests" / "host_linear_solver.cpp"
    hdr = ROOT / "rdcfes_amd" / "host" / "rdc_host.h"
    if not out.exists() or out.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime):
        subprocess.run(["g++", "-O2", "-std=c++17", str(src), "-o", str(out)], check=True)
    return out


def _solve(solver, d, rp, col, val, b, tol):
    rp.astype(np.int64).tofile(d / "row_ptr.bin")
    col.astype(np.int32).tofile(d / "col_idx.bin")
    val.astype(np.float64).tofile(d / "val.bin")
    b.astype(np.float64).tofile(d / "b.bin")
    r = subprocess.run([str(solver), str(d), repr(tol)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return np.fromfile(d / "x.bin"), int(r.stdout)


def test_solid_jacobian_with_penalty_sides(oracle, solver, tmp_path):
    mesh = gmsh.read_msh2(G / "solid_uniaxial_compression_cube.msh")
    setup = inputs.read_solid_input(G / "solid_uniaxial_compression_input.dat")
    em, mats = setup.material_table(mesh.subdomain)
    Xu = mesh.xyz
    fibre = np.tile([0.0, 0.0, 1.0], (mesh.conn.shape[0], 1))
    rp, col, val, rhs = oracle.assemble(oracle.MODEL_SOLID, 8, mesh.conn, Xu, 3, setup.params(0.1), xyz_undeformed=Xu,
                                       elem_fibre=fibre, elem_material=em, materials=mats, sides=setup.sides(mesh))
    x, its = _solve(solver, tmp_path, rp, col, val, -rhs, 1e-12)
    import scipy.sparse as sps
    A = sps.csr_matrix((val, col, rp), shape=(rhs.size, rhs.size))
    assert np.linalg.norm(A @ x + rhs) <= 1e-10 * np.linalg.norm(rhs)
    assert 0 < its < 2000


def test_reaction_diffusion_matrix(oracle, solver, tmp_path):
    conn, xyz = synth.hex_mesh(6, jitter=0.1, order="random")
    u = synth.hcc_fields(xyz)
    p = hcc_params_from_dict(synth.hcc_param_dict("full"))
    rp, col, val, rhs = oracle.assemble(oracle.MODEL_HCC, 8, conn, xyz, 3, p, u_old=u)
    x, its = _solve(solver, tmp_path, rp, col, val, rhs, 1e-13)
    import scipy.sparse as sps
    A = sps.csr_matrix((val, col, rp), shape=(rhs.size, rhs.size))
    assert np.linalg.norm(A @ x - rhs) <= 1e-11 * np.linalg.norm(rhs)
    assert its < 200
