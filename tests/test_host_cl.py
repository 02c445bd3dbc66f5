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
    st = _build(shim, 8, conn, xyz.shape[0], xyz.shape[0], (1, 8, 8, 243))
    assert st["n_wg"] == xyz.shape[0] and st["largest"] == 1
