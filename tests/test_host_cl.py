"""Cluster lists of the fused HEX8 solid kernel (rdc_prep_cl.cpp) on the CPU: structural check through the host shim
(tests/host_shim.cpp::shim_cl_build): every owned node in exactly one cluster, every (owned node, element) pair listed
exactly once with the local element / local row / node index / column slots of the mesh, limits respected; and the
property the pair order is chosen for (no node twice among the 16 lanes of an LDS atomic pass)."""
import ctypes as C

import numpy as np
import pytest

from rdcfes_amd import synth


def _build(shim, nen, conn, n_node, n_owned, lim, order=1):
    conn = np.ascontiguousarray(conn, dtype=np.uint32)
    rc = shim.shim_prep_build(nen, C.c_int64(conn.shape[0]), C.c_int64(n_node), C.c_int64(n_owned),
                              conn.ctypes.data_as(C.POINTER(C.c_uint32)), 3, C.c_int64(60 * 1024), 256)
    assert rc == 0, shim.shim_prep_error()
    st = (C.c_int64 * 8)()
    rc = shim.shim_cl_build(*lim, order, st)
    assert rc == 0, (rc, shim.shim_prep_error())
    return dict(zip(("n_wg", "n_visits", "n_pairs", "max_row", "covered", "largest", "groups", "groups_twice"), list(st)))


@pytest.mark.parametrize("lim", [(24, 192, 64, 6198), (48, 384, 128, 12398)])
@pytest.mark.parametrize("order", ["lex", "random"])
@pytest.mark.parametrize("pair_order", [0, 1])
def test_cluster_lists_hex(shim, lim, order, pair_order):
    conn, xyz = synth.hex_mesh(9, jitter=0.1, order=order)
    st = _build(shim, 8, conn, xyz.shape[0], xyz.shape[0], lim, pair_order)
    assert st["covered"] == xyz.shape[0]
    assert st["n_pairs"] == 8 * conn.shape[0]
    assert st["largest"] <= lim[0]
    # an element is evaluated once per cluster that owns one of its nodes: between once and eight times
    assert conn.shape[0] <= st["n_visits"] <= st["n_pairs"]
    assert st["n_visits"] / conn.shape[0] < 4.0
    # colour-sorted element-major order: the elements of a 16-lane group are (mostly) node-disjoint
    if pair_order == 1 and order == "lex":
        assert st["groups_twice"] < 0.5 * st["groups"]


def test_cluster_lists_on_a_ghosted_partition(shim):
    conn, xyz = synth.hex_mesh(8, jitter=0.1, order="random")
    n_owned = int(0.6 * xyz.shape[0])          # nodes >= n_owned are ghosts: their rows are not assembled
    st = _build(shim, 8, conn, xyz.shape[0], n_owned, (24, 192, 64, 6198))
    assert st["covered"] == n_owned
    owned_pairs = int((conn < n_owned).sum())
    assert st["n_pairs"] == owned_pairs


def test_cluster_lists_tiny_limits(shim):
    """limits so tight that clusters are single nodes still give complete, consistent lists"""
    conn, xyz = synth.hex_mesh(4, jitter=0.0)
    st = _build(shim, 8, conn, xyz.shape[0], xyz.shape[0], (1, 8, 8, 244))   # 27 blocks x 9 + the phase double
    assert st["n_wg"] == xyz.shape[0] and st["largest"] == 1


# ---- the HEX8 reaction-diffusion cluster kernel (rdc_hex8_cl.h) replayed on the host against the oracle ----------------------
def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _replay(shim, model, p, conn, xyz, u, aux, tracts, n_owned, lim=(24, 192, 64, 6198), order=1):
    _build(shim, 8, conn, xyz.shape[0], n_owned, lim, order)
    bptr = np.empty(shim.shim_prep_size(0), dtype=np.int64)
    shim.shim_prep_copy(0, bptr.ctypes.data_as(C.c_void_p))
    val = np.full(9 * bptr[n_owned], np.nan)
    rhs = np.full(3 * n_owned, np.nan)
    keep = [np.ascontiguousarray(a, dtype=np.float64) if a is not None else None for a in (xyz, u, aux, tracts)]
    ptr = [a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None for a in keep]
    rc = shim.shim_cl_assemble(model, C.byref(p), ptr[0], ptr[1], ptr[2], ptr[3], val.ctypes.data_as(C.POINTER(C.c_double)),
                               rhs.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 0, rc
    return val, rhs


@pytest.mark.parametrize("order", ["lex", "random"])
@pytest.mark.parametrize("params", ["full", "shipped"])
def test_hex8_cluster_replay_hcc(oracle, shim, order, params):
    from rdcfes_amd import hcc_params_from_dict
    conn, xyz = synth.hex_mesh(6, jitter=0.15, order=order)
    u = synth.hcc_fields(xyz)
    p = hcc_params_from_dict(synth.hcc_param_dict(params))
    _, _, val0, rhs0 = oracle.assemble(oracle.MODEL_HCC, 8, conn, xyz, 3, p, u_old=u)
    val, rhs = _replay(shim, 2, p, conn, xyz, u, None, None, xyz.shape[0])
    assert np.isfinite(val).all() and np.isfinite(rhs).all()
    assert _rel(val, val0) < 1e-10 and _rel(rhs, rhs0) < 1e-10
    if params == "shipped":      # every rate zero: the mass-only instantiation gives the same numbers
        val2, rhs2 = _replay(shim, 8, p, conn, xyz, u, None, None, xyz.shape[0])
        assert _rel(val2, val0) < 1e-10 and _rel(rhs2, rhs0) < 1e-10


def test_hex8_cluster_replay_ripf_and_adpm(oracle, shim):
    from rdcfes_amd import adpm_params_from_dict, ripf_params_from_dict
    conn, xyz = synth.hex_mesh(5, jitter=0.15, order="random")
    u, aux = synth.ripf_fields(xyz)
    p = ripf_params_from_dict(synth.ripf_param_dict("full"))
    _, _, val0, rhs0 = oracle.assemble(oracle.MODEL_RIPF, 8, conn, xyz, 3, p, u_old=u, aux=aux)
    val, rhs = _replay(shim, 1, p, conn, xyz, u, aux, None, xyz.shape[0])
    assert _rel(val, val0) < 1e-10 and _rel(rhs, rhs0) < 1e-10
    u, tracts = synth.adpm_fields(xyz, conn.shape[0])
    p = adpm_params_from_dict(synth.adpm_param_dict("full"), time=3.0)
    _, _, val0, rhs0 = oracle.assemble(oracle.MODEL_ADPM, 8, conn, xyz, 3, p, u_old=u, elem_fibre=tracts)
    val, rhs = _replay(shim, 4, p, conn, xyz, u, None, tracts, xyz.shape[0])
    assert _rel(val, val0) < 1e-10 and _rel(rhs, rhs0) < 1e-10


def test_hex8_cluster_replay_on_a_ghosted_partition(oracle, shim):
    from rdcfes_amd import hcc_params_from_dict
    conn, xyz = synth.hex_mesh(6, jitter=0.1, order="random")
    n_owned = int(0.55 * xyz.shape[0])
    u = synth.hcc_fields(xyz)
    p = hcc_params_from_dict(synth.hcc_param_dict("full"))
    _, _, val0, rhs0 = oracle.assemble(oracle.MODEL_HCC, 8, conn, xyz, 3, p, u_old=u, n_owned=n_owned)
    val, rhs = _replay(shim, 2, p, conn, xyz, u, None, None, n_owned)
    assert _rel(val, val0) < 1e-10 and _rel(rhs, rhs0) < 1e-10


@pytest.mark.parametrize("pair_order", [0, 1])
def test_cluster_lists_respect_interior_nodes(shim, pair_order):
    """two-part assembly of the HEX8 cluster kernels (config 5 across GPUs): with "interior_nodes" = n the clusters never mix
    interior nodes (< n) with the others, the interior clusters come first in the lists (part 1 = a leading sub-range), every
    structural property of the lists holds as before, and no row below the reported part-1 bound belongs to a later cluster.
    The node numbering is what partition.build_local produces for a rank of a 2-way split (owned nodes interior-first)."""
    from rdcfes_amd import partition
    conn, xyz = synth.hex_mesh(9, jitter=0.1, order="random")
    part = partition.partition_rcb(xyz[conn].mean(axis=1), 2)
    lp = partition.build_local(conn, xyz, part, 0, 2)
    assert 0 < lp.n_interior < lp.n_owned
    shim.shim_cl_set_interior(C.c_int64(lp.n_interior))
    try:
        st = _build(shim, 8, lp.conn, lp.xyz.shape[0], lp.n_owned, (24, 192, 64, 6198), pair_order)
        out = (C.c_int64 * 3)()
        assert shim.shim_cl_interior_stats(out) == 0
    finally:
        shim.shim_cl_set_interior(C.c_int64(-1))
    assert st["covered"] == lp.n_owned
    n_wg_int, part1_nodes, mixed = list(out)
    assert mixed == 0 and 0 < n_wg_int < st["n_wg"]
    assert 0 < part1_nodes <= lp.n_interior
    # the split costs little: at most a few clusters more than without it
    st0 = _build(shim, 8, lp.conn, lp.xyz.shape[0], lp.n_owned, (24, 192, 64, 6198), pair_order)
    assert st["n_wg"] <= 1.15 * st0["n_wg"] + 2
