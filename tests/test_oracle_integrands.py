"""Known-answer tests for the oracle's integrand restatements (SURVEY §8c KATs 2-6).
PARITY UNPINNED: the reference ships no expected outputs, so these closed forms / consistency
identities are what pins oracle/rdc_oracle.c."""
import numpy as np
import pytest

from rdcfes_amd import (SolidMaterial, hcc_params_from_dict, pihna_params_from_dict, ripf_params_from_dict, synth)

TET = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], float)
HEX = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], float)


def _mass_stiff(oracle, nen, X):
    phi, dphi, jxw = oracle.fe_reinit(nen, X)
    M = np.einsum("q,qi,qj->ij", jxw, phi, phi)
    K = np.einsum("q,qid,qjd->ij", jxw, dphi, dphi)
    return M, K, phi, dphi, jxw


def _blk(Ke, nen, a, b):
    return Ke[a * nen:(a + 1) * nen, b * nen:(b + 1) * nen]


@pytest.mark.parametrize("nen,X", [(4, TET * [1.0, 2.0, 0.5]), (8, HEX * [1.0, 2.0, 0.5])])
def test_pihna_uniform_shipped_state(oracle, nen, X):
    """run/PIHNA input: (n,c,h,v,a) = (0,0,0,7170,0) everywhere -> Ve = 1, Ua = 0, Tau = (1-Te)^3."""
    d = synth.pihna_param_dict("shipped")
    p = pihna_params_from_dict(d)
    u = np.tile([0, 0, 0, 7170.0, 0], (nen, 1))
    Ke, Fe = oracle.element(oracle.MODEL_PIHNA, nen, X, u, p)
    M, K, *_ = _mass_stiff(oracle, nen, X)
    dt2 = d["time_step"] / 2
    Te = 7170.0 / d["cells_max_capacity"]
    Tau = (1 - Te) ** 3
    dT = -3.0 / d["cells_max_capacity"] * (1 - Te) ** 2
    Ka = d["cytokines_max_capacity"]
    nec_v = d["necrosis/v"] / d["cells_max_capacity"]
    v = 7170.0
    rs = M.sum(axis=1)
    np.testing.assert_allclose(Fe[3 * nen:4 * nen], v * rs, rtol=1e-13)
    for a in (0, 1, 2, 4):
        np.testing.assert_allclose(Fe[a * nen:(a + 1) * nen], 0.0, atol=1e-12)
    np.testing.assert_allclose(_blk(Ke, nen, 0, 0), (1 - dt2 * nec_v * v) * M, rtol=1e-13)
    np.testing.assert_allclose(_blk(Ke, nen, 3, 3), M + dt2 * d["diffuse/v"] * Tau * K, rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(_blk(Ke, nen, 3, 4), -dt2 * d["produce/v"] * Tau * (1 / Ka) * v * M, rtol=1e-12)
    np.testing.assert_allclose(_blk(Ke, nen, 3, 0), -dt2 * (0.0 - nec_v * v) * M + dt2 * 0 * M, rtol=1e-12)
    # c row: Ve = 1 -> c2h(1-Ve) = 0, h2c*Ve*h with h = 0; produce_c*Tau stays
    np.testing.assert_allclose(_blk(Ke, nen, 1, 1), (1 - dt2 * (d["produce/c"] * Tau)) * M, rtol=1e-12)
    np.testing.assert_allclose(_blk(Ke, nen, 1, 2), -dt2 * d["switch/h/to/c"] * M, rtol=1e-12)
    np.testing.assert_allclose(_blk(Ke, nen, 4, 4), (1 + dt2 * d["decay/a"]) * M, rtol=1e-12)
    for a, b in ((0, 4), (4, 0), (1, 4), (2, 4)):
        assert np.all(_blk(Ke, nen, a, b) == 0.0)
    assert dT < 0


@pytest.mark.parametrize("nen,X", [(4, TET), (8, HEX)])
def test_zero_rate_limit_is_mass_matrix(oracle, nen, X):
    rng = np.random.default_rng(3)
    X = X + 0.05 * rng.standard_normal(X.shape)
    M, *_ = _mass_stiff(oracle, nen, X)
    p = pihna_params_from_dict({"time_step": 0.1, "cells_max_capacity": 10.0, "cells_max_capacity/exponent": 3.0})
    u = rng.uniform(0.1, 1.0, (nen, 5))
    Ke, Fe = oracle.element(oracle.MODEL_PIHNA, nen, X, u, p)
    for a in range(5):
        np.testing.assert_allclose(Fe[a * nen:(a + 1) * nen], M @ u[:, a], rtol=1e-13)
        for b in range(5):
            np.testing.assert_allclose(_blk(Ke, nen, a, b), M if a == b else 0 * M, rtol=1e-13, atol=1e-18)


def _fd_jacobian(f, u, h):
    J = np.empty((f(u).size, u.size))
    flat = u.ravel()
    for k in range(flat.size):
        step = h * max(1.0, abs(flat[k]))
        up, um = flat.copy(), flat.copy()
        up[k] += step
        um[k] -= step
        J[:, k] = (f(up.reshape(u.shape)) - f(um.reshape(u.shape))) / (2 * step)
    return J


@pytest.mark.parametrize("nen,X", [(4, TET), (8, HEX)])
def test_pihna_matrix_is_linearisation_of_rhs(oracle, nen, X):
    """Fe = M u + (dt/2) f(u), Ke = M - (dt/2) df/du  =>  Ke + dFe/du = 2 M (block-diagonal), for every
    block except [3][3], where the reference omits d/dv of produce_v*Tau*Ua*v's explicit v
    (src/pihna.C:708-718 has no produce_v*Tau*Ua*phi_j*phi_i term) -- reproduced, and pinned here."""
    rng = np.random.default_rng(4)
    X = X + 0.05 * rng.standard_normal(X.shape)
    d = synth.pihna_param_dict("full")
    d.update({"cells_max_capacity": 100.0, "cells_min_capacity": 0.5, "cytokines_max_capacity": 2.0,
              "necrosis/c": 5.0, "necrosis/h": 2.0, "necrosis/v": 3.0, "decay/a": 0.7, "uptake/a/from/v": 0.01,
              "secrete/a/from/c": 0.02, "secrete/a/from/h": 0.03})
    p = pihna_params_from_dict(d)
    u = rng.uniform(2.0, 8.0, (nen, 5))  # all > Lambda_k, Te in (0,1), Ve in (0,1)
    Ke, Fe = oracle.element(oracle.MODEL_PIHNA, nen, X, u, p)
    M, K, phi, dphi, jxw = _mass_stiff(oracle, nen, X)
    # dFe/du in the oracle's var-major dof order
    f = lambda uu: oracle.element(oracle.MODEL_PIHNA, nen, X, uu, p)[1]
    Jn = _fd_jacobian(f, u, 1e-6)  # columns ordered (node, var)
    J = np.empty_like(Jn)
    for j in range(nen):
        for b in range(5):
            J[:, b * nen + j] = Jn[:, j * 5 + b]
    S = Ke + J
    # expected defect of block [3][3]
    uq = phi @ u
    Te = uq[:, :4].sum(axis=1) / d["cells_max_capacity"]
    Tau = (1 - Te) ** 3
    Ua = uq[:, 4] / (uq[:, 4] + d["cytokines_max_capacity"])
    defect = d["time_step"] / 2 * d["produce/v"] * np.einsum("q,qi,qj->ij", jxw * Tau * Ua, phi, phi)
    for a in range(5):
        for b in range(5):
            expect = 2 * M if a == b else 0 * M
            if (a, b) == (3, 3):
                expect = expect + defect
            np.testing.assert_allclose(_blk(S, nen, a, b), expect, rtol=2e-6, atol=2e-7 * np.abs(Ke).max())


@pytest.mark.parametrize("nen,X", [(4, TET), (8, HEX)])
def test_ripf_matrix_is_linearisation_of_rhs(oracle, nen, X):
    rng = np.random.default_rng(5)
    X = X + 0.05 * rng.standard_normal(X.shape)
    p = ripf_params_from_dict(synth.ripf_param_dict("full"))
    u = np.stack([rng.uniform(-500, -100, nen), rng.uniform(0.05, 0.2, nen), rng.uniform(0.08, 0.2, nen)], axis=1)
    aux = np.stack([rng.choice([-0.5, 0.5], nen) * 0 + 0.5, np.full(nen, -0.5), rng.uniform(5, 60, nen)], axis=1)
    Ke, Fe = oracle.element(oracle.MODEL_RIPF, nen, X, u, p, aux=aux)
    M, *_ = _mass_stiff(oracle, nen, X)
    f = lambda uu: oracle.element(oracle.MODEL_RIPF, nen, X, uu, p, aux=aux)[1]
    Jn = _fd_jacobian(f, u, 1e-6)
    J = np.empty_like(Jn)
    for j in range(nen):
        for b in range(3):
            J[:, b * nen + j] = Jn[:, j * 3 + b]
    S = Ke + J
    for a in range(3):
        for b in range(3):
            np.testing.assert_allclose(_blk(S, nen, a, b), 2 * M if a == b else 0 * M, rtol=2e-6,
                                       atol=2e-7 * np.abs(Ke).max())
    assert np.abs(_blk(Ke, nen, 1, 0)).max() == 0.0  # never touched, src/ripf.C:599-662


def test_rigid_motion_invariance(oracle):
    rng = np.random.default_rng(6)
    p = pihna_params_from_dict(synth.pihna_param_dict("full"))
    X = TET + 0.1 * rng.standard_normal(TET.shape)
    u = np.column_stack([rng.uniform(0, 500, 4), rng.uniform(0, 2e3, 4), rng.uniform(0, 2e3, 4),
                         rng.uniform(3e3, 7e3, 4), rng.uniform(0, 1e-8, 4)])
    Q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    if np.linalg.det(Q) < 0:
        Q[:, 0] *= -1
    Ke0, Fe0 = oracle.element(oracle.MODEL_PIHNA, 4, X, u, p)
    Ke1, Fe1 = oracle.element(oracle.MODEL_PIHNA, 4, X @ Q.T + [3.0, -2.0, 5.0], u, p)
    np.testing.assert_allclose(Ke1, Ke0, rtol=1e-9, atol=1e-12 * np.abs(Ke0).max())
    np.testing.assert_allclose(Fe1, Fe0, rtol=1e-9, atol=1e-12 * np.abs(Fe0).max())


def test_hcc_reference_quirks_are_reproduced(oracle):
    """App. D.1-2: capacity term in off-diagonal blocks [0][1],[0][2],[1][0]; d/dn of the c equation
    accumulated into [1][1]; [1][2] never written."""
    rng = np.random.default_rng(7)
    X = TET + 0.05 * rng.standard_normal(TET.shape)
    M, *_ = _mass_stiff(oracle, 4, X)
    # all rates zero: only the capacity terms survive
    p = hcc_params_from_dict({"time_step": 0.01, "cells/max_capacity": 1.0, "cells/max_capacity/exponent": 3.0})
    u = rng.uniform(0, 0.3, (4, 3))
    Ke, Fe = oracle.element(oracle.MODEL_HCC, 4, X, u, p)
    expect = {(0, 0): 1, (0, 1): 1, (0, 2): 1, (1, 0): 1, (1, 1): 2, (1, 2): 0, (2, 0): 0, (2, 1): 0, (2, 2): 1}
    for (a, b), m in expect.items():
        np.testing.assert_allclose(_blk(Ke, 4, a, b), m * M, rtol=1e-13, atol=1e-18)
    p2 = hcc_params_from_dict(synth.hcc_param_dict("full"))
    Ke2, _ = oracle.element(oracle.MODEL_HCC, 4, X, u, p2)
    assert np.all(_blk(Ke2, 4, 1, 2) == 0.0)


# ---- solid ----------------------------------------------------------------------------------
def test_hyperelastic_identity_state(oracle):
    E, nu = 2.0e3, 0.4
    mu, lam = 0.5 * E / (1 + nu), E * nu / ((1 + nu) * (1 - 2 * nu))
    sig, C = oracle.hyperelastic_point(np.eye(3), [1, 1, 1], [0, 0, 1], E, nu, 0.0)
    np.testing.assert_allclose(sig, 0.0, atol=1e-10)
    Ciso = np.zeros((6, 6))
    Ciso[:3, :3] = lam
    Ciso[np.arange(3), np.arange(3)] += 2 * mu
    Ciso[np.arange(3, 6), np.arange(3, 6)] = mu
    np.testing.assert_allclose(C, Ciso, rtol=1e-12, atol=1e-9)


def test_hyperelastic_uniaxial_stretch_closed_form(oracle):
    E, nu, s = 1.0e3, 0.3, 1.3
    mu, lam = 0.5 * E / (1 + nu), E * nu / ((1 + nu) * (1 - 2 * nu))
    F = np.diag([s, 1.0, 1.0])
    sig, _ = oracle.hyperelastic_point(np.linalg.inv(F), [1, 1, 1], [0, 0, 1], E, nu, 0.0)
    J = s
    beta = J * (-mu / J + lam / 2 * J - lam / 2 / J)
    np.testing.assert_allclose(np.diag(sig), [(mu * s * s + beta) / J, (mu + beta) / J, (mu + beta) / J], rtol=1e-12)
    np.testing.assert_allclose(sig - np.diag(np.diag(sig)), 0.0, atol=1e-10)


def test_hyperelastic_tangent_closed_form_with_growth(oracle):
    """The 3^8 push-forward of hyperlastic_inline.h:100-149 equals
    (1/det F)[alpha (M M^T)_ij d_kl - beta (M_ik M_jl + M_il M_jk)], M = F Fp F^-1 (see rdc_solid.hip)."""
    rng = np.random.default_rng(8)
    E, nu, K = 2.0e3, 0.4, 50.0
    mu, lame = 0.5 * E / (1 + nu), E * nu / ((1 + nu) * (1 - 2 * nu))
    F = np.eye(3) + 0.15 * rng.standard_normal((3, 3))
    lam = np.array([1.1, 0.95, 1.2])
    fib = rng.standard_normal(3)
    sig, C = oracle.hyperelastic_point(np.linalg.inv(F), lam, fib, E, nu, K)
    detF = np.linalg.det(F)
    Je = detF / lam.prod()
    Mm = F @ np.diag(lam) @ np.linalg.inv(F)
    Q = Mm @ Mm.T
    dW = -mu / Je + lame / 2 * Je - lame / 2 / Je
    d2W = mu / Je ** 2 + lame / 2 + lame / 2 / Je ** 2
    beta, alpha = Je * dW, Je * dW + Je * Je * d2W
    a = F @ (fib / np.linalg.norm(fib))
    np.testing.assert_allclose(sig, (mu * F @ F.T + beta * Q - K * np.outer(a, a)) / detF, rtol=1e-11, atol=1e-9)
    V = [(0, 0), (1, 1), (2, 2), (0, 1), (1, 2), (0, 2)]
    Cc = np.empty((6, 6))
    for p, (i, j) in enumerate(V):
        for q, (k, l) in enumerate(V):
            Cc[p, q] = (alpha * Q[i, j] * (k == l) - beta * (Mm[i, k] * Mm[j, l] + Mm[i, l] * Mm[j, k])) / detF
    np.testing.assert_allclose(C, Cc, rtol=1e-10, atol=1e-8)


@pytest.mark.parametrize("nen,X", [(4, TET), (8, HEX)])
def test_solid_jacobian_is_derivative_of_residual_without_growth(oracle, nen, X):
    rng = np.random.default_rng(9)
    Xu = X + 0.03 * rng.standard_normal(X.shape)
    x = Xu + 0.05 * rng.standard_normal(X.shape)
    mat = SolidMaterial(2.0e3, 0.4, 0.0, (0.0, 0.0, 0.0))
    Je, Re = oracle.solid_element(nen, x, Xu, [0, 0, 1], mat, 0.3)
    f = lambda xx: oracle.solid_element(nen, xx, Xu, [0, 0, 1], mat, 0.3, request_jacobian=False)[1]
    Jn = _fd_jacobian(f, x, 1e-6)  # columns (node, dir)
    J = np.empty_like(Jn)
    for j in range(nen):
        for b in range(3):
            J[:, b * nen + j] = Jn[:, j * 3 + b]
    np.testing.assert_allclose(Je, J, rtol=1e-5, atol=1e-6 * np.abs(Je).max())


def test_solid_symmetric_fill_mirrors_upper_blocks(oracle):
    rng = np.random.default_rng(10)
    Xu = HEX + 0.03 * rng.standard_normal(HEX.shape)
    x = Xu + 0.04 * rng.standard_normal(HEX.shape)
    mat = SolidMaterial(1.0e3, 0.3, 20.0, (0.3, 0.1, 0.2))
    J0, R0 = oracle.solid_element(8, x, Xu, [1, 2, 3], mat, 0.5, use_symmetry=False)
    J1, R1 = oracle.solid_element(8, x, Xu, [1, 2, 3], mat, 0.5, use_symmetry=True)
    np.testing.assert_allclose(R0, R1, rtol=0, atol=0)
    nd = 24
    for ii in range(3):
        for jj in range(3):
            for i in range(8):
                for j in range(i, 8):
                    assert J1[ii * 8 + i, jj * 8 + j] == J0[ii * 8 + i, jj * 8 + j]
                    if i != j:
                        assert J1[ii * 8 + j, jj * 8 + i] == J0[jj * 8 + i, ii * 8 + j]


@pytest.mark.parametrize("nen,X,side", [(4, TET, 2), (8, HEX, 5)])
def test_solid_side_penalty_closed_form(oracle, nen, X, side):
    """flat side moved rigidly by d: R_i = penalty * (d - ratio*ubar) * area/nsn, free (NaN) component skipped."""
    Xu = X.copy()
    d = np.array([0.01, -0.02, 0.03])
    x = Xu + d
    disp = np.array([0.05, np.nan, -0.1])
    pt, pen = 0.4, 1.0e5
    Je, Re = oracle.solid_side(nen, side, x, Xu, disp, pt, pen)
    nodes = {(4, 2): [1, 2, 3], (8, 5): [4, 5, 6, 7]}[(nen, side)]
    area = {(4, 2): np.sqrt(3) / 2, (8, 5): 1.0}[(nen, side)]
    ratio = pt * 1.000001
    for di in range(3):
        r = Re[di * nen:(di + 1) * nen]
        if di == 1:
            assert np.all(r == 0.0)
            continue
        expect = np.zeros(nen)
        expect[nodes] = pen * (d[di] - ratio * disp[di]) * area / len(nodes)
        np.testing.assert_allclose(r, expect, rtol=1e-12, atol=1e-9)
    # Jacobian: penalty * side mass matrix on the constrained directions only
    assert np.all(Je[1 * nen:2 * nen] == 0.0)
    np.testing.assert_allclose(Je[:nen, :nen].sum(), pen * area, rtol=1e-12)
    np.testing.assert_allclose(Je[:nen, :nen], Je[2 * nen:, 2 * nen:], rtol=0, atol=0)
