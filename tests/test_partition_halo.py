"""Multi-GPU host logic on CPU: RCB partition, ghost layer, halo plan and the halo exchange itself
with world_size-2 (and 3) gloo process groups.  Also checks the design claim of SURVEY §8e: with one
ghost layer every rank assembles the complete rows of its owned nodes without any matrix exchange
(the oracle plays the kernel here; the HIP kernel is checked against the same oracle in -m gpu)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rdcfes_amd import partition, pihna_params_from_dict, synth
from rdcfes_amd.halo import HaloExchange


def test_rcb_is_balanced_and_complete():
    conn, xyz = synth.kuhn_tet_mesh(5, order="random")
    cen = xyz[conn].mean(axis=1)
    for k in (2, 3, 8):
        part = partition.partition_rcb(cen, k)
        cnt = np.bincount(part, minlength=k)
        assert cnt.sum() == conn.shape[0] and cnt.max() - cnt.min() <= 1


@pytest.mark.parametrize("nparts", [2, 4])
def test_local_partitions_cover_mesh_and_halo_plan_is_symmetric(nparts):
    conn, xyz = synth.kuhn_tet_mesh(5, order="random")
    part = partition.partition_rcb(xyz[conn].mean(axis=1), nparts)
    lps = [partition.build_local(conn, xyz, part, r, nparts) for r in range(nparts)]
    owned = np.concatenate([lp.node_global[:lp.n_owned] for lp in lps])
    assert np.array_equal(np.sort(owned), np.arange(xyz.shape[0]))  # every node owned exactly once
    assert sum(lp.n_elem_owned for lp in lps) == conn.shape[0]
    for lp in lps:
        # element list = exactly the elements touching an owned node
        g = lp.node_global[lp.conn.astype(np.int64)]
        np.testing.assert_array_equal(g, conn[lp.elem_global])
        assert np.all((lp.conn < lp.n_owned).any(axis=1))
        # interior-first numbering: an element of an interior node has no ghost node; every other owned node has one
        assert 0 <= lp.n_interior <= lp.n_owned
        has_ghost = (lp.conn >= lp.n_owned).any(axis=1)
        near = np.zeros(lp.node_global.size, dtype=bool)
        near[lp.conn[has_ghost].ravel()] = True
        assert not near[:lp.n_interior].any() and near[lp.n_interior:lp.n_owned].all()
        for q, ids in lp.recv_ids.items():
            assert np.all(ids >= lp.n_owned)
            other = lps[q]
            np.testing.assert_array_equal(lp.node_global[ids], other.node_global[other.send_ids[lp.rank]])
        ghosts = np.arange(lp.n_owned, lp.node_global.size)
        got = np.sort(np.concatenate(list(lp.recv_ids.values()))) if lp.recv_ids else np.empty(0, int)
        np.testing.assert_array_equal(got, ghosts)


def test_rows_assembled_per_rank_equal_global_rows(oracle):
    conn, xyz = synth.kuhn_tet_mesh(4, order="random")
    u = synth.pihna_fields(xyz)
    p = pihna_params_from_dict(synth.pihna_param_dict("full"))
    rp, col, val, rhs = oracle.assemble(0, 4, conn, xyz, 5, p, u_old=u)
    nparts = 3
    part = partition.partition_rcb(xyz[conn].mean(axis=1), nparts)
    for r in range(nparts):
        lp = partition.build_local(conn, xyz, part, r, nparts)
        lrp, lcol, lval, lrhs = oracle.assemble(0, 4, lp.conn, lp.xyz, 5, p, u_old=u[lp.node_global], n_owned=lp.n_owned)
        for ln in range(lp.n_owned):
            g = lp.node_global[ln]
            for a in range(5):
                lr, gr = ln * 5 + a, g * 5 + a
                assert abs(lrhs[lr] - rhs[gr]) <= 1e-12 * abs(rhs).max()
                lc = lcol[lrp[lr]:lrp[lr + 1]]
                gc = lp.node_global[lc // 5] * 5 + lc % 5
                order = np.argsort(gc)
                np.testing.assert_array_equal(gc[order], col[rp[gr]:rp[gr + 1]])
                np.testing.assert_allclose(lval[lrp[lr]:lrp[lr + 1]][order], val[rp[gr]:rp[gr + 1]], rtol=1e-12,
                                           atol=1e-14 * np.abs(val).max())


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _halo_worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        conn, xyz = synth.kuhn_tet_mesh(n, order="random")
        part = partition.partition_rcb(xyz[conn].mean(axis=1), world)
        lp = partition.build_local(conn, xyz, part, rank, world)
        nv = 5
        truth = (np.arange(xyz.shape[0] * nv, dtype=np.float64).reshape(-1, nv) + 0.25)
        u = torch.full((lp.node_global.size, nv), float("nan"), dtype=torch.float64)
        u[:lp.n_owned] = torch.from_numpy(truth[lp.node_global[:lp.n_owned]])
        hx = HaloExchange(lp, nv, "cpu")
        for step in range(2):  # twice: buffers are reusable
            hx.exchange(u)
            ok = bool(torch.equal(u, torch.from_numpy(truth[lp.node_global])))
            u[:lp.n_owned] += 1.0
            truth += 1.0
            if not ok:
                break
        dist.barrier()
        q.put((rank, ok, hx.bytes_per_step))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_halo_exchange_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_halo_worker, args=(r, world, port, 4, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert all(b > 0 for _, _, b in res)


def test_eight_way_partition_of_k20_plan_and_exchange():
    """The shape of the 8-GPU run (BASELINE cfg4) at small size: 8-way RCB of K(20), plan symmetry over all 28 rank pairs,
    and the grouped exchange itself on an 8-rank gloo group."""
    conn, xyz = synth.kuhn_tet_mesh(20, order="lex")
    part = partition.partition_rcb(xyz[conn].mean(axis=1), 8)
    owner = partition.node_owners(conn, part, xyz.shape[0], 8)
    lps = [partition.build_local(conn, xyz, part, r, 8, owner=owner) for r in range(8)]
    assert sum(lp.n_owned for lp in lps) == xyz.shape[0]
    assert sum(lp.n_elem_owned for lp in lps) == conn.shape[0]
    for lp in lps:
        assert lp.n_interior > 0.5 * lp.n_owned                      # most rows can be assembled before the halo lands
        assert lp.conn.shape[0] < 1.6 * lp.n_elem_owned               # one ghost layer, not more
        assert 3 <= len(lp.recv_ids) <= 7                             # an octant of the cube touches the 7 others at most
        for q, ids in lp.recv_ids.items():
            np.testing.assert_array_equal(lp.node_global[ids], lps[q].node_global[lps[q].send_ids[lp.rank]])
        for q, ids in lp.send_ids.items():
            assert np.all(ids < lp.n_owned) and lp.rank in lps[q].recv_ids
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_halo_worker, args=(r, 8, port, 6, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(res) == 8 and all(ok for _, ok, _ in res)


def _coord_halo_worker(rank, world, port, q):
    """cfg5 halo plan on a HEX8 partition: the HCC unknowns (3 per node) and the current coordinates of the moving mesh
    (3 per node) travel in ONE grouped message per peer (HaloExchange.exchange_many)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        conn, Xu = synth.hex_mesh(7, jitter=0.1, order="random")
        part = partition.partition_rcb(Xu[conn].mean(axis=1), world)
        lp = partition.build_local(conn, Xu, part, rank, world)
        x_true = Xu + synth.solid_displacement(Xu)          # the moved mesh
        u_true = synth.hcc_fields(Xu)
        u = torch.full((lp.node_global.size, 3), float("nan"), dtype=torch.float64)
        x = torch.full((lp.node_global.size, 3), float("nan"), dtype=torch.float64)
        u[:lp.n_owned] = torch.from_numpy(u_true[lp.node_global[:lp.n_owned]])
        x[:lp.n_owned] = torch.from_numpy(x_true[lp.node_global[:lp.n_owned]])
        hx = HaloExchange(lp, 6, "cpu")
        ok = True
        for step in range(2):
            hx.exchange_many([u, x])
            ok = ok and bool(torch.equal(u, torch.from_numpy(u_true[lp.node_global]))) and bool(torch.equal(x, torch.from_numpy(x_true[lp.node_global])))
            u[:lp.n_owned] *= 0.5; u_true = u_true * 0.5           # next step's state; the ghost rows are stale until the exchange
            x[:lp.n_owned] += 0.01; x_true = x_true + 0.01
        # the interior-first numbering holds for hexes too: no element of an interior node contains a ghost
        touches_ghost = (lp.conn >= lp.n_owned).any(axis=1)
        near = np.unique(lp.conn[touches_ghost])
        ok = ok and not np.any(near[near < lp.n_owned] < lp.n_interior) and lp.n_interior > 0
        dist.barrier()
        q.put((rank, ok, hx.bytes_per_step, len(hx._ops)))
    finally:
        dist.destroy_process_group()


def test_coordinate_halo_plan_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_coord_halo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res)
    assert all(nops == 2 for _, _, _, nops in res)          # one send + one receive per peer: both fields in one message
    assert all(b % 48 == 0 and b > 0 for _, _, b, _ in res)  # 6 doubles per interface node
