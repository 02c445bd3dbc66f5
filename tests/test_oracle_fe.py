"""Known-answer tests pinning the FE building blocks of the oracle (SURVEY §8c KAT 1).
The reference holds no fixtures for these (PARITY UNPINNED); the values below are closed forms."""
import numpy as np
import pytest


def test_tet4_partition_of_unity_and_volume(oracle):
    rng = np.random.default_rng(0)
    X = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], float) + 0.1 * rng.standard_normal((4, 3))
    phi, dphi, jxw = oracle.fe_reinit(4, X)
    assert phi.shape == (5, 4)
    np.testing.assert_allclose(phi.sum(axis=1), 1.0, atol=1e-15)
    np.testing.assert_allclose(dphi.sum(axis=1), 0.0, atol=1e-13)
    vol = np.linalg.det(X[1:] - X[0]) / 6.0
    np.testing.assert_allclose(jxw.sum(), vol, rtol=1e-14)
    # libMesh rule: centroid weight negative, four equal positive weights (App. B.2)
    assert jxw[0] < 0 and np.allclose(jxw[1:], jxw[1])
    np.testing.assert_allclose(jxw[0] / jxw.sum(), -0.8, rtol=1e-14)


def test_tet4_rule_exact_for_cubics(oracle):
    # integral over the unit tet of x^a y^b z^c = a! b! c! / (a+b+c+3)!
    from math import factorial as f
    X = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], float)
    phi, _, jxw = oracle.fe_reinit(4, X)
    pts = phi @ X
    for a in range(4):
        for b in range(4 - a):
            for c in range(4 - a - b):
                num = np.sum(jxw * pts[:, 0] ** a * pts[:, 1] ** b * pts[:, 2] ** c)
                np.testing.assert_allclose(num, f(a) * f(b) * f(c) / f(a + b + c + 3), rtol=1e-13, atol=1e-16)


def test_tet4_mass_matrix_closed_form(oracle):
    X = np.array([[0, 0, 0], [2, 0, 0], [0, 1, 0], [0, 0, 3]], float)
    phi, _, jxw = oracle.fe_reinit(4, X)
    M = np.einsum("q,qi,qj->ij", jxw, phi, phi)
    V = 1.0
    np.testing.assert_allclose(M, V / 20.0 * (np.ones((4, 4)) + np.eye(4)), rtol=1e-13)


def test_tet4_gradients_reproduce_linear_field(oracle):
    rng = np.random.default_rng(1)
    X = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], float) + 0.2 * rng.standard_normal((4, 3))
    g = np.array([0.3, -1.2, 2.0])
    vals = X @ g + 0.7
    _, dphi, _ = oracle.fe_reinit(4, X)
    for q in range(5):
        np.testing.assert_allclose(vals @ dphi[q], g, rtol=1e-12)


def test_hex8_basics(oracle):
    rng = np.random.default_rng(2)
    X = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], float)
    Xj = X * np.array([2.0, 1.5, 0.5]) + 0.05 * rng.standard_normal((8, 3))
    phi, dphi, jxw = oracle.fe_reinit(8, Xj)
    assert phi.shape == (8, 8)
    np.testing.assert_allclose(phi.sum(axis=1), 1.0, atol=1e-15)
    np.testing.assert_allclose(dphi.sum(axis=1), 0.0, atol=1e-13)
    # undistorted brick: volume and exact mass matrix row sums
    phi, dphi, jxw = oracle.fe_reinit(8, X * np.array([2.0, 1.5, 0.5]))
    np.testing.assert_allclose(jxw.sum(), 1.5, rtol=1e-14)
    M = np.einsum("q,qi,qj->ij", jxw, phi, phi)
    np.testing.assert_allclose(M.sum(axis=1), 1.5 / 8.0, rtol=1e-13)
    np.testing.assert_allclose(M[0, 0], 1.5 / 27.0, rtol=1e-13)   # (V/8) * (2/3)^3
    np.testing.assert_allclose(M[0, 6], 1.5 / 216.0, rtol=1e-13)  # (V/8) * (1/3)^3
    g = np.array([0.3, -1.2, 2.0])
    vals = (X * np.array([2.0, 1.5, 0.5])) @ g
    for q in range(8):
        np.testing.assert_allclose(vals @ dphi[q], g, rtol=1e-12)


def test_pattern_matches_dense_construction(oracle):
    from rdcfes_amd import synth
    conn, xyz = synth.kuhn_tet_mesh(3, order="random")
    n = xyz.shape[0]
    row_ptr, col, bptr, bcol = oracle.build_pattern(4, conn, n, n, 2)
    adj = np.zeros((n, n), bool)
    for e in conn:
        adj[np.ix_(e, e)] = True
    for i in range(n):
        np.testing.assert_array_equal(bcol[bptr[i]:bptr[i + 1]], np.nonzero(adj[i])[0])
    # scalar expansion: row = node*nvar + a, cols = nodecol*nvar + b ascending
    assert row_ptr[-1] == col.size == 4 * adj.sum()
    r = 2 * 5 + 1
    expect = (np.nonzero(adj[5])[0][:, None] * 2 + np.arange(2)[None, :]).ravel()
    np.testing.assert_array_equal(col[row_ptr[r]:row_ptr[r + 1]], expect)
