"""Element-visit / moment formulation of the shipped-pattern PIHNA TET4 assembly (rdc_tet4_ev.h, rdc_prep_ev.cpp) on
the CPU: the kernel's phases replayed on the host from the same work lists and the same device functions
(tests/host_shim.cpp::shim_ev_assemble) against the oracle's whole-mesh assembly."""
import ctypes as C

import numpy as np
import pytest

from rdcfes_amd import pihna_params_from_dict, synth


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _ev(shim, conn, xyz, u, p, n_owned, budget=54000, gen=0):
    conn = np.ascontiguousarray(conn, dtype=np.uint32)
    rc = shim.shim_prep_build(4, C.c_int64(conn.shape[0]), C.c_int64(xyz.shape[0]), C.c_int64(n_owned),
                              conn.ctypes.data_as(C.POINTER(C.c_uint32)), 5, C.c_int64(60 * 1024), 256)
    assert rc == 0, shim.shim_prep_error()
    stats = (C.c_int64 * 6)()
    rc = shim.shim_ev_build(C.c_int64(budget), stats)
    assert rc == 0, shim.shim_prep_error()
    n_wg, n_vis, n_rows, nls, max_out, covered = list(stats)
    assert covered == n_owned                      # every owned node is in exactly one cluster
    bptr = np.empty(shim.shim_prep_size(0), dtype=np.int64)
    shim.shim_prep_copy(0, bptr.ctypes.data_as(C.c_void_p))
    val = np.full(25 * bptr[n_owned], np.nan)
    rhs = np.full(5 * n_owned, np.nan)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    u = np.ascontiguousarray(u, dtype=np.float64)
    rc = shim.shim_ev_assemble_gen(C.byref(p), dp(xyz), dp(u), dp(val), dp(rhs), int(gen))
    assert rc == 0, rc
    return val, rhs, dict(n_wg=n_wg, n_vis=n_vis, n_rows=n_rows, nls=nls, max_out=max_out)


@pytest.mark.parametrize("order", ["lex", "random"])
@pytest.mark.parametrize("pvariant", ["shipped", "shipped_realexp"])
def test_ev_replay_matches_oracle(oracle, shim, order, pvariant):
    conn, xyz = synth.kuhn_tet_mesh(7, order=order)
    u = synth.pihna_fields(xyz)
    d = synth.pihna_param_dict("shipped")
    if pvariant == "shipped_realexp":
        d["cells_max_capacity/exponent"] = 2.5
    p = pihna_params_from_dict(d)
    _, _, val0, rhs0 = oracle.assemble(0, 4, conn, xyz, 5, p, u_old=u)
    val, rhs, st = _ev(shim, conn, xyz, u, p, xyz.shape[0])
    assert np.isfinite(val).all() and np.isfinite(rhs).all()      # every CSR value and rhs entry is produced
    assert rel(val, val0) < 1e-10 and rel(rhs, rhs0) < 1e-10
    # an element is visited by every cluster that owns one of its nodes; a visit serves 1..4 rows
    assert st["n_rows"] == 4 * conn.shape[0]
    assert conn.shape[0] <= st["n_vis"] <= st["n_rows"]
    if order == "lex":
        assert st["n_rows"] / st["n_vis"] > 1.5


@pytest.mark.parametrize("order", ["lex", "random"])
@pytest.mark.parametrize("pvariant,gen", [("full", 0), ("realexp", 0), ("shipped", 1), ("taxis_v_only", 0), ("diffuse_c_only", 0), ("taxis_h_only", 0)])
def test_ev_replay_general_parameters(oracle, shim, order, pvariant, gen):
    """Every term of the model on (22 moments, rdc_tet4_ev.h GEN), single transport terms, and the general kernel on the shipped values."""
    conn, xyz = synth.kuhn_tet_mesh(7, order=order)
    u = synth.pihna_fields(xyz)
    if pvariant in ("full", "realexp", "shipped"):
        d = synth.pihna_param_dict(pvariant)
    else:
        d = synth.pihna_param_dict("shipped")
        d.update({"taxis_v_only": {"taxis/v": 0.3, "uptake/a/from/v": 2.0e-5}, "diffuse_c_only": {"diffuse/c": 0.2},
                  "taxis_h_only": {"taxis/h": 0.15}}[pvariant])
    p = pihna_params_from_dict(d)
    _, _, val0, rhs0 = oracle.assemble(0, 4, conn, xyz, 5, p, u_old=u)
    val, rhs, st = _ev(shim, conn, xyz, u, p, xyz.shape[0], gen=gen)
    assert np.isfinite(val).all() and np.isfinite(rhs).all()
    assert rel(val, val0) < 1e-10 and rel(rhs, rhs0) < 1e-10


@pytest.mark.parametrize("pvariant", ["shipped", "full"])
def test_ev_replay_background_skip(oracle, shim, pvariant):
    """Visits in the background state (n = c = h = a = 0, v > 0 at their four vertices) skip the moments that are sums of exact
    zeros: same matrix as with everything evaluated (and as the oracle), with most visits of the synthetic state skipping."""
    conn, xyz = synth.kuhn_tet_mesh(8, order="random")
    u = synth.pihna_fields(xyz)
    bgn = (u[:, [0, 1, 2, 4]] == 0).all(1) & (u[:, 3] > 0)
    assert 0.5 < bgn[conn].all(1).mean() < 1.0                 # most elements are background, some are not
    p = pihna_params_from_dict(synth.pihna_param_dict(pvariant))
    _, _, val0, rhs0 = oracle.assemble(0, 4, conn, xyz, 5, p, u_old=u)
    shim.shim_ev_set_background(0)
    try:
        val_all, rhs_all, _ = _ev(shim, conn, xyz, u, p, xyz.shape[0])
    finally:
        shim.shim_ev_set_background(1)
    val, rhs, _ = _ev(shim, conn, xyz, u, p, xyz.shape[0])
    assert rel(val, val0) < 1e-10 and rel(rhs, rhs0) < 1e-10
    assert rel(val, val_all) < 1e-14 and rel(rhs, rhs_all) < 1e-14
    # rows of nodes all of whose elements are background: the n, c, h, a right-hand sides are exactly 0, as upstream
    quiet = np.ones(xyz.shape[0], bool)
    quiet[np.unique(conn[~bgn[conn].all(1)])] = False
    assert quiet.sum() > 100 and np.all(rhs.reshape(-1, 5)[quiet][:, [0, 1, 2, 4]] == 0.0)
    assert np.all(rhs0.reshape(-1, 5)[quiet][:, [0, 1, 2, 4]] == 0.0)


def test_ev_replay_on_a_ghosted_partition(oracle, shim):
    conn, xyz = synth.kuhn_tet_mesh(6, order="random")
    u = synth.pihna_fields(xyz)
    p = pihna_params_from_dict(synth.pihna_param_dict("shipped"))
    n_owned = int(0.6 * xyz.shape[0])
    conn = conn[(conn < n_owned).any(axis=1)]
    _, _, val0, rhs0 = oracle.assemble(0, 4, conn, xyz, 5, p, u_old=u, n_owned=n_owned)
    val, rhs, st = _ev(shim, conn, xyz, u, p, n_owned)
    assert rel(val, val0) < 1e-10 and rel(rhs, rhs0) < 1e-10
    assert st["n_rows"] == int((conn < n_owned).sum())


def test_ev_replay_clamped_branches(oracle, shim):
    """saturated / empty crowding, vascular fraction at 0 and 1, sub-threshold vasculature, all-zero nodes (0/0 -> NaN in the
    same entries as the oracle): the states of test_gpu_parity.py::test_pihna_shipped_pattern_branches"""
    conn, xyz = synth.kuhn_tet_mesh(6, order="random")
    rng = np.random.default_rng(5)
    u = synth.pihna_fields(xyz)
    p = pihna_params_from_dict(synth.pihna_param_dict("shipped"))
    x = xyz[:, 0]
    u[x < 0.25, :4] *= 60.0
    m = (x >= 0.25) & (x < 0.4)
    u[m, 3] = rng.uniform(0.0, 2.0 * p.cells_min_capacity, int(m.sum()))
    u[(x >= 0.4) & (x < 0.5), 1:3] = 0.0
    u[(x >= 0.5) & (x < 0.6), 3] = 0.0
    u[x > 0.7] = 0.0
    _, _, val0, rhs0 = oracle.assemble(0, 4, conn, xyz, 5, p, u_old=u)
    val, rhs, _ = _ev(shim, conn, xyz, u, p, xyz.shape[0])
    assert np.isnan(val0).any()
    assert np.array_equal(np.isnan(val), np.isnan(val0)) and np.array_equal(np.isnan(rhs), np.isnan(rhs0))
    ok, okr = ~np.isnan(val0), ~np.isnan(rhs0)
    assert rel(val[ok], val0[ok]) < 1e-10 and rel(rhs[okr], rhs0[okr]) < 1e-10


# ---- coefficient-form element visits (rdc_tet4_evc.h) for the three-unknown models ------------------------------------------
def _evc(shim, model, p, conn, xyz, u, aux, n_owned, budget=54000):
    conn = np.ascontiguousarray(conn, dtype=np.uint32)
    rc = shim.shim_prep_build(4, C.c_int64(conn.shape[0]), C.c_int64(xyz.shape[0]), C.c_int64(n_owned),
                              conn.ctypes.data_as(C.POINTER(C.c_uint32)), 3, C.c_int64(60 * 1024), 256)
    assert rc == 0, shim.shim_prep_error()
    stats = (C.c_int64 * 6)()
    rc = shim.shim_ev_build(C.c_int64(budget), stats)
    assert rc == 0, shim.shim_prep_error()
    assert stats[5] == n_owned
    bptr = np.empty(shim.shim_prep_size(0), dtype=np.int64)
    shim.shim_prep_copy(0, bptr.ctypes.data_as(C.c_void_p))
    val = np.full(9 * bptr[n_owned], np.nan)
    rhs = np.full(3 * n_owned, np.nan)
    keep = [np.ascontiguousarray(a, dtype=np.float64) if a is not None else None for a in (xyz, u, aux)]
    ptr = [a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None for a in keep]
    rc = shim.shim_evc_assemble(model, C.byref(p), ptr[0], ptr[1], ptr[2], val.ctypes.data_as(C.POINTER(C.c_double)),
                                rhs.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 0, rc
    return val, rhs


@pytest.mark.parametrize("order", ["lex", "random"])
@pytest.mark.parametrize("params", ["full", "shipped"])
def test_evc_replay_ripf(oracle, shim, order, params):
    from rdcfes_amd import ripf_params_from_dict
    conn, xyz = synth.kuhn_tet_mesh(6, order=order)
    u, aux = synth.ripf_fields(xyz)
    p = ripf_params_from_dict(synth.ripf_param_dict(params))
    _, _, val0, rhs0 = oracle.assemble(oracle.MODEL_RIPF, 4, conn, xyz, 3, p, u_old=u, aux=aux)
    val, rhs = _evc(shim, 1, p, conn, xyz, u, aux, xyz.shape[0])
    assert np.isfinite(val).all() and np.isfinite(rhs).all()
    assert rel(val, val0) < 1e-10 and rel(rhs, rhs0) < 1e-10
    if params == "shipped":   # the reduced instantiation of the shipped parameter pattern gives the same numbers
        val2, rhs2 = _evc(shim, 7, p, conn, xyz, u, aux, xyz.shape[0])
        assert rel(val2, val0) < 1e-10 and rel(rhs2, rhs0) < 1e-10


@pytest.mark.parametrize("params", ["full", "shipped"])
def test_evc_replay_hcc_on_a_ghosted_partition(oracle, shim, params):
    from rdcfes_amd import hcc_params_from_dict
    conn, xyz = synth.kuhn_tet_mesh(6, order="random")
    n_owned = int(0.6 * xyz.shape[0])
    conn = conn[(conn < n_owned).any(axis=1)]
    u = synth.hcc_fields(xyz)
    p = hcc_params_from_dict(synth.hcc_param_dict(params))
    _, _, val0, rhs0 = oracle.assemble(oracle.MODEL_HCC, 4, conn, xyz, 3, p, u_old=u, n_owned=n_owned)
    val, rhs = _evc(shim, 2 if params == "full" else 8, p, conn, xyz, u, None, n_owned)
    assert rel(val, val0) < 1e-10 and rel(rhs, rhs0) < 1e-10
