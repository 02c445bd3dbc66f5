"""BASELINE cfg5 as a SEQUENCE through the C++ host mirror (rdcfes_amd/host/rdc_host.h): the call order of the
reference's coupled driver, src/coupled_hcc.C:98-130 -- reaction-diffusion step on the current mesh, then on loading
steps `SolidSystem::run_solver()` (Newton: one rdc_solid_assemble per iteration), `post_process()`, `update_data()` --
on the reference's own shipped cube (tests/golden/solid_uniaxial_compression_cube.msh, its BC sets 0 and 5).
Every assembled object the driver hands to its solvers is checked against the oracle on the same state."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

from rdcfes_amd import SolidMaterial, SolidParams, gmsh, hcc_params_from_dict, inputs, synth

ROOT = Path(__file__).resolve().parent.parent
G = ROOT / "tests" / "golden"
pytestmark = pytest.mark.gpu
TOL = 1e-10


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope="module")
def driver():
    from rdcfes_amd import build
    lib = build.build(verbose=False)
    out = ROOT / "tests" / "_build" / "solid_mirror_driver"
    out.parent.mkdir(exist_ok=True)
    src = ROOT / "tests" / "solid_mirror_driver.cpp"
    hdr = ROOT / "rdcfes_amd" / "host" / "rdc_host.h"
    if not out.exists() or out.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime, lib.stat().st_mtime):
        subprocess.run(["g++", "-O2", "-std=c++17", str(src), "-o", str(out), f"-L{lib.parent}", "-lrdc_assembly",
                        f"-Wl,-rpath,{lib.parent}"], check=True)
    return out


def _params_txt(setup, extra):
    """es.parameters as input() of src/coupled_hcc.C:144-373 types them"""
    L = []
    for bc, disp in sorted(setup.bcs.items()):
        L.append(f"point BC/{bc}/displacement " + " ".join("nan" if np.isnan(v) else repr(float(v)) for v in disp))
    L.append("string BCs " + " ".join(str(b) for b in sorted(setup.bcs)))
    L.append(f"real BCs/displacement_penalty {setup.penalty!r}")
    for m, mat in setup.materials.items():
        h = f"material/{m}/Hyperelastic/"
        L += [f"real {h}Young {mat.Young!r}", f"real {h}Poisson {mat.Poisson!r}", f"real {h}FibreStiffness {mat.FibreStiffness!r}"]
        L += [f"real {h}VolumetricStretchRatio/rate_{d} {mat.rate[d]!r}" for d in range(3)]
    for k, (t, v) in extra.items():
        L.append(f"{t} {k} {v}")
    return "\n".join(L) + "\n"


def test_cfg5_call_order_on_the_shipped_cube(oracle, driver, tmp_path):
    mesh = gmsh.read_msh2(G / "solid_uniaxial_compression_cube.msh")
    kv = inputs.parse_getpot((G / "solid_uniaxial_compression_input.dat").read_text())
    # the shipped file's material keys are not the ones the code reads: give the read keys values so
    # that growth and the fibre term take part, as in run/Coupled/HCC/input.dat:44-53
    kv.update({"material/0/Hyperelastic/Young": "2.0e3", "material/0/Hyperelastic/Poisson": "0.4",
               "material/0/Hyperelastic/FibreStiffness": "30.0",
               "material/0/Hyperelastic/VolumetricStretchRatio/rate_0": "0.3",
               "material/0/Hyperelastic/VolumetricStretchRatio/rate_1": "0.2",
               "material/0/Hyperelastic/VolumetricStretchRatio/rate_2": "0.1"})
    setup = inputs.SolidSetup(kv)
    assert setup.loading_step == 0.1 and sorted(setup.bcs) == [0, 5]
    conn, Xu = mesh.conn, mesh.xyz
    ne, nn = conn.shape[0], Xu.shape[0]
    fibre = np.tile([0.0, 0.6, 0.8], (ne, 1)) + 0.05 * np.random.default_rng(3).standard_normal((ne, 3))
    u0 = synth.hcc_fields(Xu)
    hcc_d = synth.hcc_param_dict("full")
    p_hcc = hcc_params_from_dict(hcc_d)
    from rdcfes_amd.params import HCC_DEFAULTS
    n_steps, ltp = 4, [2, 4]
    extra = {k: ("real", repr(float(v))) for k, v in {**HCC_DEFAULTS, **hcc_d}.items()}
    extra.update({
        "loading_step": ("real", repr(setup.loading_step)), "number_of_time_steps": ("int", n_steps),
        "loading_time_points": ("string", " ".join(map(str, ltp))), "test/with_rd": ("bool", "true"),
        "solver/quiet": ("bool", "true"), "solver/nonlinear/max_nonlinear_iterations": ("int", 15),
        "solver/nonlinear/relative_step_tolerance": ("real", "1e-10"), "solver/nonlinear/relative_residual_tolerance": ("real", "1e-9"),
        "solver/nonlinear/absolute_residual_tolerance": ("real", "1e-9"), "solver/nonlinear/require_reduction": ("bool", "false"),
        "solver/linear/max_linear_iterations": ("int", 50000), "solver/linear/initial_linear_tolerance": ("real", "1e-12"),
    })
    conn.astype(np.uint32).tofile(tmp_path / "conn.bin")
    Xu.astype(np.float64).tofile(tmp_path / "xyz.bin")
    mesh.subdomain.astype(np.int32).tofile(tmp_path / "subdomain.bin")
    sides = []
    for bid in np.unique(mesh.face_tag):
        e, s = mesh.sides_with_boundary_id(int(bid))
        sides += [(int(a), int(b), int(bid)) for a, b in zip(e, s)]
    np.array(sides, dtype=np.int64).tofile(tmp_path / "sides.bin")
    fibre.astype(np.float64).tofile(tmp_path / "fibre.bin")
    u0.astype(np.float64).tofile(tmp_path / "u.bin")
    (tmp_path / "params.txt").write_text(_params_txt(setup, extra))
    r = subprocess.run([str(driver), str(tmp_path), "8"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    log = (tmp_path / "log.txt").read_text().splitlines()

    em, mats = setup.material_table(mesh.subdomain)
    osides = setup.sides(mesh)
    rp = np.fromfile(tmp_path / "row_ptr.bin", dtype=np.int64)
    col = np.fromfile(tmp_path / "col_idx.bin", dtype=np.int32)
    rd = lambda name, dt=np.float64: np.fromfile(tmp_path / name, dtype=dt)
    import scipy.sparse as sps
    import scipy.sparse.linalg as spla
    x_prev = Xu.copy()
    for t in range(1, n_steps + 1):
        # ---- reaction-diffusion step: assembled on the mesh as the last loading step left it (src/coupled_hcc.C:111-114)
        xyz_t = rd(f"rd_xyz_{t}.bin").reshape(nn, 3)
        assert np.array_equal(xyz_t, x_prev)
        u_old = rd(f"rd_old_{t}.bin").reshape(nn, 3)
        rp0, col0, val0, rhs0 = oracle.assemble(oracle.MODEL_HCC, 8, conn, xyz_t, 3, p_hcc, u_old=u_old)
        assert np.array_equal(rp, rp0) and np.array_equal(col, col0)
        assert rel(rd(f"rd_val_{t}.bin"), val0) < TOL and rel(rd(f"rd_rhs_{t}.bin"), rhs0) < TOL
        sol = spla.spsolve(sps.csr_matrix((val0, col0, rp0), shape=(3 * nn, 3 * nn)).tocsc(), rhs0)
        assert rel(rd(f"rd_sol_{t}.bin"), np.maximum(sol, 0.0)) < 1e-8          # check_solution: clamp of the solved state
        if t not in ltp:
            continue
        # ---- loading step: first Newton assembly against the oracle ...
        pt = setup.loading_step * (ltp.index(t) + 1)
        sp = setup.params(pt)
        x0 = rd(f"sb_xyz_{t}.bin").reshape(nn, 3)
        assert np.array_equal(x0, x_prev)
        _, _, jv, jr = oracle.assemble(oracle.MODEL_SOLID, 8, conn, x0, 3, sp, xyz_undeformed=Xu, elem_fibre=fibre,
                                       elem_material=em, materials=mats, sides=osides)
        assert rel(rd(f"sb_val_{t}.bin"), jv) < TOL and rel(rd(f"sb_rhs_{t}.bin"), jr) < TOL
        # ... the converged state is an equilibrium of the oracle's residual ...
        (line,) = [ln for ln in log if ln.startswith(f"step {t} pseudo_time")]
        f = dict(zip(line.split()[2::2], line.split()[3::2]))
        assert int(f["converged"]) == 1 and 1 <= int(f["newton_iterations"]) <= 15
        assert abs(float(f["pseudo_time"]) - pt) < 1e-15
        x1 = rd(f"sb_sol_{t}.bin").reshape(nn, 3)
        _, _, _, r1 = oracle.assemble(oracle.MODEL_SOLID, 8, conn, x1, 3, sp, xyz_undeformed=Xu, elem_fibre=fibre,
                                      elem_material=em, materials=mats, sides=osides, request_jacobian=False)
        assert np.linalg.norm(r1) < 1e-7 * np.linalg.norm(jr)
        assert abs(np.linalg.norm(r1) - float(f["last_residual"])) <= 1e-6 * np.linalg.norm(jr)
        # ... that honours the penalty boundary conditions (side set 0 fixed, side set 5 pushed down by ratio * 0.75) ...
        disp = rd(f"sb_disp_{t}.bin").reshape(nn, 3)
        assert np.allclose(disp, x1 - Xu, rtol=0, atol=1e-14)
        n0 = np.unique(mesh.face_nodes[mesh.face_tag == 0])
        n5 = np.unique(mesh.face_nodes[mesh.face_tag == 5])
        assert np.abs(disp[n0]).max() < 1e-3
        assert np.abs(disp[n5, 2] + 0.75 * pt * 1.000001).max() < 1e-3
        assert np.abs(disp).max() > 0.02                                         # the mesh did move
        # ... and post_process() ran on it
        pr0, vm0, fc0 = oracle.solid_post_process(8, conn, x1, Xu, fibre, em, mats, pt)
        assert rel(rd(f"sb_press_{t}.bin"), pr0) < TOL and rel(rd(f"sb_vm_{t}.bin"), vm0) < TOL
        fs = rd(f"sb_fibre_{t}.bin").reshape(ne, 6)
        assert np.array_equal(fs[:, :3], fibre) and rel(fs[:, 3:], fc0) < TOL
        x_prev = x1
    assert np.array_equal(rd("aux_old.bin").reshape(nn, 3), Xu)                 # update_data(): aux old <- current (undeformed)
