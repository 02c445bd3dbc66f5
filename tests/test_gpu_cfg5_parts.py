"""BASELINE config 5 (coupled_hcc + solid_system on the deforming HEX8 mesh) across partitions.

  * two-part assembly of the HEX8 cluster kernels: the cluster lists respect "interior_nodes" (interior clusters first),
    part 1 + part 2 == whole call for k_hex8_cl (HCC) and k_solid_cl (+ the penalty sides, which part 2 adds behind part 1);
  * two ranks on the one GPU (gloo): RCB partition of a hex mesh, ONE grouped halo carrying the HCC unknowns and the current
    coordinates of the moved mesh (src/solid_system.C:103-123 moves the mesh; src/coupled_hcc.C:98-130 call order), part 1
    assembled while the ghost values are still NaN, the rows of both systems against the oracle's global assembly."""
import numpy as np
import pytest

from rdcfes_amd import AssemblyContext, SolidMaterial, SolidParams, hcc_params_from_dict, synth
from rdcfes_amd.context import FIELD_ELEM_FIBRE, FIELD_OLD_SOLUTION, FIELD_UNDEFORMED_XYZ

pytestmark = pytest.mark.gpu
TOL = 1e-10


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _case(n=8):
    rng = np.random.default_rng(3)
    conn, Xu = synth.hex_mesh(n, jitter=0.1, order="lex")
    x = Xu + synth.solid_displacement(Xu, amp=0.02)
    em = (np.linalg.norm(Xu[conn].mean(axis=1) - 0.5, axis=1) < 0.3).astype(np.int32)
    mats = [SolidMaterial(2.0e3, 0.4, 0.0, (0.0, 0.0, 0.0)), SolidMaterial(1.5e3, 0.35, 40.0, (0.3, 0.2, 0.1))]
    fibre = rng.standard_normal((conn.shape[0], 3))
    se0, ss0 = synth.boundary_sides(8, conn, Xu, 2, 0.0)
    se1, ss1 = synth.boundary_sides(8, conn, Xu, 2, 1.0)
    sides = (np.concatenate([se0, se1]), np.concatenate([ss0, ss1]),
             np.concatenate([np.zeros((se0.size, 3)), np.tile([np.nan, np.nan, -0.75], (se1.size, 1))]))
    return conn, Xu, x, em, mats, fibre, sides


@pytest.mark.parametrize("frac", [0.0, 0.45, 1.0])
def test_two_part_hex8_cluster_kernels_equal_whole(frac):
    conn, Xu, x, em, mats, fibre, sides = _case()
    nn = Xu.shape[0]
    u = synth.hcc_fields(Xu)
    ph = hcc_params_from_dict(synth.hcc_param_dict("full"))
    sp = SolidParams(0.4, 1.0e5, 0, 0)
    # "interior" = a prefix of the (lexicographic) node numbering none of whose elements reaches beyond it... for this
    # test ANY prefix works: there are no ghosts, part 1 just has to stay inside it and part 2 do the rest
    n_int = int(frac * nn)
    import torch
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for system in ("hcc", "solid"):
        with AssemblyContext(0) as ctx:
            ctx.set_option("interior_nodes", n_int)
            ctx.mesh_upload(8, conn, x, 3)
            if system == "hcc":
                ctx.field_upload(FIELD_OLD_SOLUTION, u)
                whole = lambda: ctx.assemble_hcc(ph)
                part = lambda k, s: ctx.assemble_hcc_part(ph, k, s.cuda_stream)
            else:
                ctx.field_upload(FIELD_UNDEFORMED_XYZ, Xu)
                ctx.field_upload(FIELD_ELEM_FIBRE, fibre)
                ctx.solid_set_materials(em, mats)
                ctx.solid_set_sides(*sides)
                whole = lambda: ctx.solid_assemble(sp, True)
                part = lambda k, s: ctx.solid_assemble_part(sp, True, k, s.cuda_stream)
            whole()
            val0, rhs0 = ctx.csr_download()
            rp, _ = ctx.csr_pattern()
            # overwrite every row, then the two parts on two streams (part 2 concurrently with part 1: the sides wait inside)
            ctx.mesh_update_coords(Xu)
            whole()
            val1, rhs1 = ctx.csr_download()
            assert not np.array_equal(val1, val0)
            ctx.mesh_update_coords(x)
            ctx.synchronize()
            part(1, s1)
            s1.synchronize()
            n1 = ctx.part1_nodes()
            vala, rhsa = ctx.csr_download()
            assert 0 <= n1 <= n_int
            if frac > 0.3:
                assert n1 > 0
            # part 1 completes rows [0, n1) -- except for the penalty-side contributions, which belong to part 2 -- and touches nothing at or beyond n_int
            assert np.array_equal(vala[rp[3 * n_int]:], val1[rp[3 * n_int]:]) and np.array_equal(rhsa[3 * n_int:], rhs1[3 * n_int:])
            if system == "hcc" and n1 > 0:
                assert rel(vala[:rp[3 * n1]], val0[:rp[3 * n1]]) < 1e-13
            part(1, s1)                       # again, now racing with part 2 on the other stream
            part(2, s2)
            s1.synchronize(); s2.synchronize()
            val, rhs = ctx.csr_download()
        assert rel(val, val0) < 1e-13 and rel(rhs, rhs0) < 1e-13, system


def _two_rank_cfg5(rank, world, port, q):
    import os
    import torch
    import torch.distributed as dist
    from rdcfes_amd import partition
    from rdcfes_amd.halo import HaloExchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        conn, Xu, x, em, mats, fibre, _ = _case(8)
        u = synth.hcc_fields(Xu)
        ph = hcc_params_from_dict(synth.hcc_param_dict("full"))
        sp = SolidParams(0.4, 1.0e5, 0, 0)
        part = partition.partition_rcb(Xu[conn].mean(axis=1), world)
        lp = partition.build_local(conn, Xu, part, rank, world)
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        ng = lp.node_global
        u_l = torch.full((ng.size, 3), float("nan"), dtype=torch.float64, device=dev)
        u_l[:lp.n_owned] = torch.from_numpy(u[ng[:lp.n_owned]]).to(dev)
        x_start = x[ng].copy()
        x_start[lp.n_owned:] = np.nan                                   # the ghosts' current positions arrive by halo
        hx = HaloExchange(lp, 6, dev)
        main = torch.cuda.current_stream()
        with AssemblyContext(0) as hcc, AssemblyContext(0) as sol:
            for c in (hcc, sol):
                c.set_stream(main.cuda_stream)
                c.set_option("interior_nodes", int(lp.n_interior))
                c.mesh_upload(8, lp.conn, x_start, 3, n_owned=lp.n_owned)
            hcc.field_bind_device(FIELD_OLD_SOLUTION, u_l.data_ptr(), u_l.numel())
            sol.field_upload(FIELD_UNDEFORMED_XYZ, Xu[ng])
            sol.field_upload(FIELD_ELEM_FIBRE, fibre[lp.elem_global])
            sol.solid_set_materials(em[lp.elem_global], mats)
            x_sol, x_hcc = sol.coords_tensor(), hcc.coords_tensor()
            # part 1 of both systems: interior rows, ghost unknowns and ghost coordinates still NaN
            hcc.assemble_hcc_part(ph, 1, main.cuda_stream)
            sol.solid_assemble_part(sp, True, 1, main.cuda_stream)
            main.synchronize()
            hx.exchange_many([u_l, x_sol])                             # one message per peer: unknowns + coordinates
            x_hcc[lp.n_owned:].copy_(x_sol[lp.n_owned:])              # the HCC system runs on the same moved mesh
            hcc.assemble_hcc_part(ph, 2, main.cuda_stream)
            sol.solid_assemble_part(sp, True, 2, main.cuda_stream)
            res = {}
            for name, c in (("hcc", hcc), ("solid", sol)):
                val, rhs = c.csr_download()
                assert np.isfinite(val).all() and np.isfinite(rhs).all(), name
                rp, col = c.csr_pattern()
                res[name] = (val, rhs, rp, col)
            n1 = (hcc.part1_nodes(), sol.part1_nodes())
        # reference: the global assemblies restricted to this rank's rows
        ref = {"hcc": O.assemble(O.MODEL_HCC, 8, conn, x, 3, ph, u_old=u),
               "solid": O.assemble(O.MODEL_SOLID, 8, conn, x, 3, sp, xyz_undeformed=Xu, elem_fibre=fibre, elem_material=em,
                                   materials=mats, request_jacobian=True)}
        err = 0.0
        for name in ("hcc", "solid"):
            val, rhs, rp, col = res[name]
            grp, gcol, gval, grhs = ref[name]
            for ln in range(0, lp.n_owned, 5):
                g = ng[ln]
                for a in range(3):
                    lr, gr = ln * 3 + a, g * 3 + a
                    lc = col[rp[lr]:rp[lr + 1]]
                    gc = ng[lc // 3] * 3 + lc % 3
                    order = np.argsort(gc)
                    assert np.array_equal(gc[order], gcol[grp[gr]:grp[gr + 1]])
                    err = max(err, np.abs(val[rp[lr]:rp[lr + 1]][order] - gval[grp[gr]:grp[gr + 1]]).max() / np.abs(gval).max())
                    err = max(err, abs(rhs[lr] - grhs[gr]) / np.abs(grhs).max())
        dist.barrier()
        q.put((rank, err, n1, int(lp.n_interior)))
    finally:
        dist.destroy_process_group()


def test_two_rank_cfg5_hcc_on_moved_mesh_and_solid_rows():
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_cfg5, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(err < TOL for _, err, _, _ in res), res
    assert all(n1[0] > 0 and n1[1] > 0 for _, _, n1, _ in res), res      # both systems did assemble rows in part 1
