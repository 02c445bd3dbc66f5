"""RIPF check_solution (src/ripf.C:675-775): SURVEY §8(f) rank 1, the post-solve nodal bookkeeping."""
import numpy as np
import pytest

from rdcfes_amd import RipfCheckParams, synth


def _state(n, seed=0):
    rng = np.random.default_rng(seed)
    sol = np.column_stack([rng.uniform(-1500, 1500, n), rng.uniform(-0.2, 1.2, n), rng.uniform(-0.1, 0.5, n)])
    prev = np.column_stack([rng.uniform(-1000, 1000, n), rng.uniform(0, 1, n), rng.uniform(0, 0.3, n)])
    rt = np.column_stack([rng.uniform(0, 67, n), rng.uniform(0, 6.7, n), np.full(n, -7.0)])
    return sol, prev, rt


@pytest.mark.parametrize("day,expect", [(0, lambda b, f: b / 28 * 1), (27, lambda b, f: b / 28 * 28),
                                         (28, lambda b, f: f / 8 * 1 + b), (35, lambda b, f: f / 8 * 8 + b),
                                         (36, lambda b, f: b + f), (400, lambda b, f: b + f)])
def test_oracle_known_answers(oracle, day, expect):
    """Shipped schedule of run/RIPF133/input.dat: 28 broad + 8 focus fractions."""
    sol, prev, rt = _state(50)
    p = RipfCheckParams(0.1, -1000.0, 1000.0, 28, 8, day, 0)
    s, pv, td, r, aux, mx = oracle.ripf_check_solution(p, sol, prev, rt)
    np.testing.assert_allclose(r[:, 2], expect(rt[:, 0], rt[:, 1]), rtol=1e-15)
    assert mx == r[:, 2].max()
    np.testing.assert_array_equal(s[:, 0], np.clip(sol[:, 0], -1000.0, 1000.0))
    np.testing.assert_array_equal(s[:, 1:], np.maximum(sol[:, 1:], 0.0))
    np.testing.assert_array_equal(pv, sol)                       # prev_soln = soln: the UNCLAMPED state (:769)
    np.testing.assert_allclose(td, (s - prev) * (1.0 / 0.1), rtol=1e-15)
    np.testing.assert_array_equal(aux, np.column_stack([td[:, 1], td[:, 2], r[:, 2]]))


def test_oracle_max_floor(oracle):
    """RT_total_max starts at -1 (:705): all-negative doses leave it there (upstream then aborts, :772)."""
    sol, prev, rt = _state(10)
    rt[:, :2] = -5.0
    *_, mx = oracle.ripf_check_solution(RipfCheckParams(0.1, -1000.0, 1000.0, 28, 8, 50, 0), sol, prev, rt)
    assert mx == -1.0


@pytest.mark.gpu
@pytest.mark.parametrize("day", [3, 30, 40])
def test_gpu_matches_oracle(oracle, day):
    from rdcfes_amd import (AssemblyContext, FIELD_AUX_NODAL, FIELD_OLD_SOLUTION, FIELD_PREV_SOLUTION, FIELD_RT_DOSE,
                            FIELD_TIME_DERIV, ripf_params_from_dict)
    conn, xyz = synth.kuhn_tet_mesh(12, order="random")
    n = xyz.shape[0]
    sol, prev, rt = _state(n, seed=day)
    # asymmetric clamp bounds: with the shipped +-1000 three nodes at +1000 and one at -1000 put HU = 0 (up to the
    # rounding of the interpolation) on a quadrature point, exactly on the jump of d(Lombda)/dHU at HU = 0
    # (src/ripf.C:532-545) -- any two evaluation orders then legitimately pick different branches
    p = RipfCheckParams(0.1, -1000.0, 1094.0, 28, 8, day, 0)
    s0, pv0, td0, r0, aux0, mx0 = oracle.ripf_check_solution(p, sol, prev, rt)
    with AssemblyContext(0) as ctx:
        ctx.mesh_upload(4, conn, xyz, 3)
        with pytest.raises(Exception):
            ctx.ripf_check_solution(p)                       # fields not set
        ctx.field_upload(FIELD_OLD_SOLUTION, sol)
        ctx.field_upload(FIELD_PREV_SOLUTION, prev)
        ctx.field_upload(FIELD_RT_DOSE, rt)
        mx = ctx.ripf_check_solution(p)
        got = [ctx.field_download(f, 3 * n).reshape(n, 3) for f in
               (FIELD_OLD_SOLUTION, FIELD_PREV_SOLUTION, FIELD_TIME_DERIV, FIELD_RT_DOSE, FIELD_AUX_NODAL)]
        # the aux record it leaves is what the next assembly reads
        rp = ripf_params_from_dict(synth.ripf_param_dict("shipped"))
        rp.RT_dose_total_max = int(mx)                       # upstream stores it truncated to int (:771)
        ctx.assemble_ripf(rp)
        val, rhs = ctx.csr_download()
    # clamp and copy are exact; the rates and the schedule are FP64 arithmetic (the device contracts a*b+c into an
    # fma, the oracle is built with -ffp-contract=off): a few ulp
    np.testing.assert_array_equal(got[0], s0)
    np.testing.assert_array_equal(got[1], pv0)
    for a, b in zip(got[2:], (td0, r0, aux0)):
        np.testing.assert_allclose(a, b, rtol=1e-15, atol=0.0)
    assert abs(mx - mx0) <= 1e-15 * abs(mx0)
    _, _, val0, rhs0 = oracle.assemble(oracle.MODEL_RIPF, 4, conn, xyz, 3, rp, u_old=got[0], aux=got[4])
    assert np.linalg.norm(rhs - rhs0) <= 1e-10 * np.linalg.norm(rhs0)
    assert np.linalg.norm(val - val0) <= 1e-10 * np.linalg.norm(val0)
