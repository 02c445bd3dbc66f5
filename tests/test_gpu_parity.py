"""GPU parity: HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs.
Tolerance: north_star asks for the global residual L2 within 1e-10 relative; the same bound is
applied to the matrix (Frobenius) and to A*x for a seeded random x."""
import numpy as np
import pytest

from rdcfes_amd import (AssemblyContext, RdcError, SolidMaterial, SolidParams, hcc_params_from_dict,
                        pihna_params_from_dict, ripf_params_from_dict, synth)
from rdcfes_amd.context import (FIELD_AUX_NODAL, FIELD_ELEM_FIBRE, FIELD_OLD_SOLUTION, FIELD_UNDEFORMED_XYZ,
                                SCATTER_COLOURED, SCATTER_ROWGATHER, VARIANT_AUTO, VARIANT_GENERIC)

pytestmark = pytest.mark.gpu
TOL = 1e-10


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def csr_matvec(row_ptr, col, val, x):
    y = np.add.reduceat(val * x[col], row_ptr[:-1])
    return y


def _inputs(model, nen, n, order="random", variant="full"):
    conn, xyz = synth.kuhn_tet_mesh(n, order=order) if nen == 4 else synth.hex_mesh(n, jitter=0.15, order=order)
    aux = None
    if model == 0:
        p, u = pihna_params_from_dict(synth.pihna_param_dict(variant)), synth.pihna_fields(xyz)
    elif model == 1:
        p = ripf_params_from_dict(synth.ripf_param_dict(variant))
        u, aux = synth.ripf_fields(xyz)
    else:
        p, u = hcc_params_from_dict(synth.hcc_param_dict(variant)), synth.hcc_fields(xyz)
    return conn, xyz, u, aux, p


def _gpu_assemble(model, nen, conn, xyz, u, aux, p, strategy, variant, n_owned=None, options=()):
    nv = 5 if model == 0 else 3
    with AssemblyContext(0) as ctx:
        for key, value in options:
            ctx.set_option(key, value)
        ctx.mesh_upload(nen, conn, xyz, nv, n_owned=n_owned)
        ctx.set_scatter(strategy)
        ctx.set_kernel_variant(variant)
        ctx.field_upload(FIELD_OLD_SOLUTION, u)
        if aux is not None:
            ctx.field_upload(FIELD_AUX_NODAL, aux)
        [ctx.assemble_pihna, ctx.assemble_ripf, ctx.assemble_hcc][model](p)
        val, rhs = ctx.csr_download()
        row_ptr, col = ctx.csr_pattern()
        # a second call must give the same answer (no state carried in val/rhs between steps)
        [ctx.assemble_pihna, ctx.assemble_ripf, ctx.assemble_hcc][model](p)
        val2, rhs2 = ctx.csr_download()
    assert rel(val2, val) < 1e-13 and rel(rhs2, rhs) < 1e-13
    return row_ptr, col, val, rhs


COMBOS = [(m, nen, s, v, pv)
          for m, pvs in ((0, ("full", "shipped", "realexp")), (1, ("full", "shipped")), (2, ("full", "shipped")))
          for pv in pvs
          for nen in (4, 8)
          for s in (SCATTER_COLOURED, SCATTER_ROWGATHER)
          for v in ((VARIANT_AUTO, VARIANT_GENERIC) if nen == 4 else (VARIANT_GENERIC,))
          if not (m == 0 and nen == 8 and pv != "full")]


@pytest.mark.parametrize("model,nen,strategy,variant,pvariant", COMBOS)
def test_parity_small_mesh(oracle, model, nen, strategy, variant, pvariant):
    conn, xyz, u, aux, p = _inputs(model, nen, 6 if nen == 4 else 7, variant=pvariant)
    nv = 5 if model == 0 else 3
    rp0, col0, val0, rhs0 = oracle.assemble(model, nen, conn, xyz, nv, p, u_old=u, aux=aux)
    rp, col, val, rhs = _gpu_assemble(model, nen, conn, xyz, u, aux, p, strategy, variant)
    np.testing.assert_array_equal(rp, rp0)
    np.testing.assert_array_equal(col, col0)
    assert rel(rhs, rhs0) < TOL
    assert rel(val, val0) < TOL
    x = np.random.default_rng(1).standard_normal(xyz.shape[0] * nv)
    assert rel(csr_matvec(rp, col, val, x), csr_matvec(rp0, col0, val0, x)) < TOL
    if nen == 8 and nv == 3 and strategy == SCATTER_ROWGATHER:
        # the default above is the producer / consumer cluster kernel (rdc_hex8_cl.h); its persistent form, the other pair order
        # of the cluster lists and the pair kernels stay covered
        for options in ((("hex_kernel", 2),), (("solid_cl_order", 1),), (("hex_kernel", 1),)):
            _, _, val, rhs = _gpu_assemble(model, nen, conn, xyz, u, aux, p, strategy, variant, options=options)
            assert rel(rhs, rhs0) < TOL and rel(val, val0) < TOL, options
    if nen == 8 and nv == 5 and strategy == SCATTER_ROWGATHER:
        # five unknowns: the default above is the cluster kernel one equation row at a time (k_hex8_cl_rows); the pair kernels stay covered
        _, _, val, rhs = _gpu_assemble(model, nen, conn, xyz, u, aux, p, strategy, variant, options=(("hex_kernel", 1),))
        assert rel(rhs, rhs0) < TOL and rel(val, val0) < TOL


def test_hex8_cluster_kernel_on_a_ghosted_partition(oracle):
    """the HEX8 cluster kernel with ghost nodes: rows only for the owned nodes, every (owned node, element) pair once"""
    conn, xyz, u, aux, p = _inputs(2, 8, 7, variant="full")
    n_owned = int(0.55 * xyz.shape[0])
    conn = conn[(conn < n_owned).any(axis=1)]
    rp0, col0, val0, rhs0 = oracle.assemble(2, 8, conn, xyz, 3, p, u_old=u, n_owned=n_owned)
    rp, col, val, rhs = _gpu_assemble(2, 8, conn, xyz, u, None, p, SCATTER_ROWGATHER, VARIANT_GENERIC, n_owned=n_owned)
    np.testing.assert_array_equal(rp, rp0)
    assert rel(rhs, rhs0) < TOL and rel(val, val0) < TOL


@pytest.mark.parametrize("strategy", [SCATTER_COLOURED, SCATTER_ROWGATHER])
def test_parity_ghosted_partition(oracle, strategy):
    """rows only for owned nodes; ghost nodes contribute columns."""
    conn, xyz, u, aux, p = _inputs(0, 4, 6)
    n_node = xyz.shape[0]
    n_owned = int(0.55 * n_node)
    conn = conn[(conn < n_owned).any(axis=1)]
    rp0, col0, val0, rhs0 = oracle.assemble(0, 4, conn, xyz, 5, p, u_old=u, n_owned=n_owned)
    rp, col, val, rhs = _gpu_assemble(0, 4, conn, xyz, u, None, p, strategy, VARIANT_AUTO, n_owned=n_owned)
    assert rhs.size == n_owned * 5
    np.testing.assert_array_equal(rp, rp0)
    np.testing.assert_array_equal(col, col0)
    assert rel(rhs, rhs0) < TOL and rel(val, val0) < TOL


@pytest.mark.parametrize("pvariant", ["shipped", "realexp_shipped"])
def test_resident_element_visit_kernel_on_a_ghosted_partition(oracle, pvariant):
    """k_tet4_evq (the default of large whole-mesh launches; "ev_resident" = 2 runs it at any size): rows of owned nodes only, clusters
    handed out by the counter over several launches in a row (the record pack kernel resets it), shipped pattern with the integer and
    a real crowding exponent."""
    conn, xyz = synth.kuhn_tet_mesh(9, order="random")
    u = synth.pihna_fields(xyz)
    d = synth.pihna_param_dict("shipped")
    if pvariant == "realexp_shipped":
        d["cells_max_capacity/exponent"] = 2.5
    p = pihna_params_from_dict(d)
    n_owned = int(0.6 * xyz.shape[0])
    conn = conn[(conn < n_owned).any(axis=1)]
    _, _, val0, rhs0 = oracle.assemble(0, 4, conn, xyz, 5, p, u_old=u, n_owned=n_owned)
    with AssemblyContext(0) as ctx:
        ctx.set_option("ev_resident", 2)
        ctx.mesh_upload(4, conn, xyz, 5, n_owned=n_owned)
        ctx.field_upload(FIELD_OLD_SOLUTION, u)
        for _ in range(3):
            ctx.field_upload(FIELD_OLD_SOLUTION, 0.25 * u)
            ctx.assemble_pihna(p)
            ctx.field_upload(FIELD_OLD_SOLUTION, u)
            ctx.assemble_pihna(p)
            val, rhs = ctx.csr_download()
            assert rel(rhs, rhs0) < TOL and rel(val, val0) < TOL


def test_parity_cfg2_full_size(oracle):
    """BASELINE configs[1]: PIHNA, K(55) = 998,250 TET4 / 175,616 nodes, shipped parameters."""
    conn, xyz, u, aux, p = _inputs(0, 4, 55, order="lex", variant="shipped")
    assert conn.shape[0] == 998250 and xyz.shape[0] == 175616
    rp0, col0, val0, rhs0 = oracle.assemble(0, 4, conn, xyz, 5, p, u_old=u)
    rp, col, val, rhs = _gpu_assemble(0, 4, conn, xyz, u, None, p, SCATTER_ROWGATHER, VARIANT_AUTO)
    assert rel(rhs, rhs0) < TOL and rel(val, val0) < TOL
    rp, col, val, rhs = _gpu_assemble(0, 4, conn, xyz, u, None, p, SCATTER_COLOURED, VARIANT_AUTO)
    assert rel(rhs, rhs0) < TOL and rel(val, val0) < TOL


def test_full_size_properties_10m_tets():
    """K(119), 10.1M tets (the metric's mesh): size-independent checks instead of the oracle.
    (1) zero-rate parameters: every diagonal block is the mass matrix -> its entries sum to the
        volume (1), rhs sums to the integral of u_old; off-diagonal blocks vanish;
    (2) the two independent scatter paths (coloured RMW vs row gather) agree."""
    import torch
    conn, xyz = synth.kuhn_tet_mesh(119, order="lex")
    assert conn.shape[0] == 10110954
    u = synth.pihna_fields(xyz)
    p0 = pihna_params_from_dict({"time_step": 0.1, "cells_max_capacity": 2.39e5, "cells_max_capacity/exponent": 3.0,
                                 "cytokines_max_capacity": 1e-8})
    p1 = pihna_params_from_dict(synth.pihna_param_dict("full"))
    with AssemblyContext(0) as ctx:
        ctx.mesh_upload(4, conn, xyz, 5)
        ctx.field_upload(FIELD_OLD_SOLUTION, u)
        n_rows, nnz = ctx.csr_dims()
        ctx.set_scatter(SCATTER_ROWGATHER)
        ctx.assemble_pihna(p0)
        val, rhs = ctx.csr_download()
        row_ptr, col = ctx.csr_pattern()
        a_of_row = np.arange(n_rows) % 5
        b_of_col = col % 5
        row_of = np.repeat(np.arange(n_rows), np.diff(row_ptr))
        diag = a_of_row[row_of] == b_of_col
        assert np.abs(val[~diag]).max() == 0.0
        for a in range(5):
            s = val[diag & (a_of_row[row_of] == a)].sum()
            assert abs(s - 1.0) < 1e-9
        X = xyz[conn]
        vol = np.abs(np.einsum("ei,ei->e", X[:, 1] - X[:, 0], np.cross(X[:, 2] - X[:, 0], X[:, 3] - X[:, 0]))) / 6
        for a in range(5):
            integral = (vol * u[conn][:, :, a].mean(axis=1)).sum()
            assert abs(rhs[a::5].sum() - integral) <= 1e-9 * max(abs(integral), 1e-300)
        del val, rhs, row_of, diag, b_of_col, col
        ctx.assemble_pihna(p1)
        v_rg, r_rg = ctx.csr_download()
        ctx.set_scatter(SCATTER_COLOURED)
        ctx.assemble_pihna(p1)
        v_c, r_c = ctx.csr_download()
    assert rel(v_c, v_rg) < 1e-12 and rel(r_c, r_rg) < 1e-12


def test_moving_mesh_hcc(oracle):
    """assemble_hcc runs on the CURRENT (deformed) coordinates (src/coupled_hcc.C:98-114)."""
    conn, xyz, u, aux, p = _inputs(2, 8, 6, order="lex")
    xyz2 = xyz + synth.solid_displacement(xyz)
    _, _, val0, rhs0 = oracle.assemble(2, 8, conn, xyz2, 3, p, u_old=u)
    with AssemblyContext(0) as ctx:
        ctx.mesh_upload(8, conn, xyz, 3)
        ctx.field_upload(FIELD_OLD_SOLUTION, u)
        ctx.assemble_hcc(p)
        v_undeformed, _ = ctx.csr_download()
        ctx.mesh_update_coords(xyz2)
        ctx.assemble_hcc(p)
        val, rhs = ctx.csr_download()
    assert rel(val, val0) < TOL and rel(rhs, rhs0) < TOL
    assert rel(v_undeformed, val0) > 1e-4


def _solid_case(nen, n, seed=0):
    rng = np.random.default_rng(seed)
    conn, Xu = synth.kuhn_tet_mesh(n, jitter=0.1, order="random") if nen == 4 else synth.hex_mesh(n, jitter=0.1, order="random")
    x = Xu + synth.solid_displacement(Xu, amp=0.02)
    ne = conn.shape[0]
    cen = Xu[conn].mean(axis=1)
    em = (np.linalg.norm(cen - 0.5, axis=1) < 0.3).astype(np.int32)
    mats = [SolidMaterial(2.0e3, 0.4, 0.0, (0.0, 0.0, 0.0)), SolidMaterial(1.5e3, 0.35, 40.0, (0.3, 0.2, 0.1))]
    fibre = rng.standard_normal((ne, 3))
    se0, ss0 = synth.boundary_sides(nen, conn, Xu, 2, 0.0)
    se1, ss1 = synth.boundary_sides(nen, conn, Xu, 2, 1.0)
    se = np.concatenate([se0, se1])
    ss = np.concatenate([ss0, ss1])
    sd = np.concatenate([np.zeros((se0.size, 3)), np.tile([np.nan, np.nan, -0.75], (se1.size, 1))])
    return conn, Xu, x, em, mats, fibre, (se, ss, sd)


@pytest.mark.parametrize("nen,n", [(8, 5), (4, 4)])
@pytest.mark.parametrize("use_symmetry", [0, 1])
@pytest.mark.parametrize("jac", [True, False])
@pytest.mark.parametrize("solid_kernel,solid_gather,solid_split,cl_waves",
                         [(0, 0, 0, 31), (0, 0, 0, 62), (2, 0, 0, 31), (2, 1, 1, 31), (1, 0, 0, 31)])
def test_solid_parity(oracle, nen, n, use_symmetry, jac, solid_kernel, solid_gather, solid_split, cl_waves):
    conn, Xu, x, em, mats, fibre, sides = _solid_case(nen, n)
    sp = SolidParams(0.4, 1.0e5, use_symmetry, 0)
    _, _, val0, rhs0 = oracle.assemble(oracle.MODEL_SOLID, nen, conn, x, 3, sp, xyz_undeformed=Xu, elem_fibre=fibre,
                                       elem_material=em, materials=mats, request_jacobian=jac, sides=sides)
    with AssemblyContext(0) as ctx:
        ctx.set_option("solid_kernel", solid_kernel)   # 0 = default (fused cluster kernel for HEX8 tangents, else two-pass), 1 = coloured, 2 = two-pass
        ctx.set_option("solid_cl_waves", cl_waves)     # fused kernel: 31 = 3 consumer + 1 producer waves, 62 = 6 + 2
        ctx.set_option("solid_gather", solid_gather)   # pass 2: 0 = stores staged through LDS (default), 1 = direct
        ctx.set_option("solid_split", solid_split)     # pass 1: 0 = HEX8 row split over two threads (default), 1 = not
        ctx.mesh_upload(nen, conn, x, 3)
        ctx.field_upload(FIELD_UNDEFORMED_XYZ, Xu)
        ctx.field_upload(FIELD_ELEM_FIBRE, fibre)
        ctx.solid_set_materials(em, mats)
        ctx.solid_set_sides(*sides)
        ctx.solid_assemble(sp, jac)
        val, rhs = ctx.csr_download()
        ctx.solid_assemble(sp, jac)                    # repeatable (buffers reused), and deterministic
        val2, rhs2 = ctx.csr_download()
    assert rel(rhs, rhs0) < TOL
    if jac:
        assert rel(val, val0) < TOL
    else:
        assert np.all(val == 0.0)
    if solid_kernel == 2 and sides[0].size == 0:
        assert np.array_equal(val, val2) and np.array_equal(rhs, rhs2)
    else:
        assert rel(rhs2, rhs0) < TOL


@pytest.mark.parametrize("solid_kernel", [0, 2])
def test_solid_parity_on_a_ghosted_partition(oracle, solid_kernel):
    """SolidSystem rows of the owned nodes only (one rank of a partition): the cluster lists of the fused kernel hold owned
    nodes, its producers read ghost nodes' coordinates"""
    conn, Xu, x, em, mats, fibre, _ = _solid_case(8, 6)
    n_owned = int(0.55 * Xu.shape[0])
    keep = (conn < n_owned).any(axis=1)
    conn, em, fibre = conn[keep], em[keep], fibre[keep]
    sp = SolidParams(0.4, 1.0e5, 0, 0)
    _, _, val0, rhs0 = oracle.assemble(oracle.MODEL_SOLID, 8, conn, x, 3, sp, xyz_undeformed=Xu, elem_fibre=fibre,
                                       elem_material=em, materials=mats, request_jacobian=True, n_owned=n_owned)
    with AssemblyContext(0) as ctx:
        ctx.set_option("solid_kernel", solid_kernel)
        ctx.mesh_upload(8, conn, x, 3, n_owned=n_owned)
        ctx.field_upload(FIELD_UNDEFORMED_XYZ, Xu)
        ctx.field_upload(FIELD_ELEM_FIBRE, fibre)
        ctx.solid_set_materials(em, mats)
        ctx.solid_assemble(sp, True)
        val, rhs = ctx.csr_download()
    assert rhs.size == 3 * n_owned
    assert rel(rhs, rhs0) < TOL and rel(val, val0) < TOL


def test_clamp_nonnegative(oracle):
    conn, xyz = synth.kuhn_tet_mesh(3)
    u = np.random.default_rng(3).standard_normal((xyz.shape[0], 5))
    with AssemblyContext(0) as ctx:
        ctx.mesh_upload(4, conn, xyz, 5)
        ctx.field_upload(FIELD_OLD_SOLUTION, u)
        ctx.clamp_nonnegative(FIELD_OLD_SOLUTION)
        out = ctx.field_download(FIELD_OLD_SOLUTION, u.size)
    np.testing.assert_array_equal(out, oracle.clamp_nonnegative(u).ravel())


def test_error_behaviour():
    """int status codes instead of libmesh_error() aborts."""
    conn, xyz = synth.kuhn_tet_mesh(2)
    p = pihna_params_from_dict(synth.pihna_param_dict())
    with AssemblyContext(0) as ctx:
        with pytest.raises(RdcError) as e:
            ctx.assemble_pihna(p)
        assert e.value.code == 3  # RDC_ERR_STATE: no mesh
        ctx.mesh_upload(4, conn, xyz, 5)
        with pytest.raises(RdcError) as e:
            ctx.assemble_pihna(p)
        assert e.value.code == 3  # old solution missing
        with pytest.raises(RdcError) as e:
            ctx.field_upload(FIELD_OLD_SOLUTION, np.zeros(7))
        assert e.value.code == 1
        with pytest.raises(RdcError) as e:
            ctx.assemble_hcc(hcc_params_from_dict(synth.hcc_param_dict()))
        assert e.value.code == 1  # nvar mismatch
        bad = conn.copy()
        bad[0, 0] = 10 ** 6
        with pytest.raises(RdcError) as e:
            ctx.mesh_upload(4, bad, xyz, 5)
        assert e.value.code == 1 and "out of range" in str(e.value)
    with pytest.raises(RdcError):
        AssemblyContext(10 ** 4)


def test_timing_hook():
    conn, xyz, u, aux, p = _inputs(0, 4, 8)
    with AssemblyContext(0) as ctx:
        ctx.mesh_upload(4, conn, xyz, 5)
        ctx.field_upload(FIELD_OLD_SOLUTION, u)
        ctx.timing_enable(True)
        ctx.assemble_pihna(p)
        ctx.synchronize()
        assert 0.0 < ctx.timing_last_ms() < 1e4
        for _ in range(4):
            ctx.assemble_pihna(p)
        samples = ctx.timing_samples_ms()          # one device time per call, oldest first; resets the pool
        assert len(samples) == 5 and all(0.0 < t < 1e4 for t in samples)
        ctx.assemble_pihna(p)
        total, n = ctx.timing_sum_ms()
        assert n == 1 and 0.0 < total < 1e4


# ---- two ranks (gloo, host-staged halo) sharing the one GPU of the test box ---------------------
def _two_rank_worker(rank, world, port, q):
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
        conn, xyz = synth.kuhn_tet_mesh(8, order="random")
        u = synth.pihna_fields(xyz)
        p = pihna_params_from_dict(synth.pihna_param_dict("full"))
        part = partition.partition_rcb(xyz[conn].mean(axis=1), world)
        lp = partition.build_local(conn, xyz, part, rank, world)
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        u_l = torch.full((lp.node_global.size, 5), float("nan"), dtype=torch.float64, device=dev)
        u_l[:lp.n_owned] = torch.from_numpy(u[lp.node_global[:lp.n_owned]]).to(dev)  # ghosts arrive by halo
        hx = HaloExchange(lp, 5, dev)
        with AssemblyContext(0) as ctx:
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            ctx.mesh_upload(4, lp.conn, lp.xyz, 5, n_owned=lp.n_owned)
            ctx.field_bind_device(FIELD_OLD_SOLUTION, u_l.data_ptr(), u_l.numel())
            # two-part assembly as bench.py overlaps it: the interior rows are assembled while the ghost values are
            # still NaN (nothing on them may read a ghost), the remaining rows after the exchange
            ctx.set_option("interior_nodes", int(lp.n_interior))
            ctx.set_option("part", 1)
            ctx.assemble_pihna(p)
            ctx.synchronize()
            hx.exchange(u_l)
            ctx.set_option("part", 2)
            ctx.assemble_pihna(p)
            val, rhs = ctx.csr_download()
            assert np.isfinite(val).all() and np.isfinite(rhs).all()
            rp, col = ctx.csr_pattern()
        # reference: the global assembly restricted to this rank's rows
        grp, gcol, gval, grhs = O.assemble(0, 4, conn, xyz, 5, p, u_old=u)
        err = 0.0
        for ln in range(0, lp.n_owned, 7):
            g = lp.node_global[ln]
            for a in range(5):
                lr, gr = ln * 5 + a, g * 5 + a
                lc = col[rp[lr]:rp[lr + 1]]
                gc = lp.node_global[lc // 5] * 5 + lc % 5
                order = np.argsort(gc)
                assert np.array_equal(gc[order], gcol[grp[gr]:grp[gr + 1]])
                ref = gval[grp[gr]:grp[gr + 1]]
                err = max(err, np.abs(val[rp[lr]:rp[lr + 1]][order] - ref).max() / np.abs(gval).max())
                err = max(err, abs(rhs[lr] - grhs[gr]) / np.abs(grhs).max())
        dist.barrier()
        q.put((rank, err))
    finally:
        dist.destroy_process_group()


def test_two_rank_partitioned_assembly():
    """N > 1 path end to end on one GPU: RCB partition, ghost layer, halo exchange into a bound
    device tensor, per-rank assembly of complete owned rows (no matrix exchange)."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(err < TOL for _, err in res), res


@pytest.mark.parametrize("opts", [{"kernel": 7}, {"kernel": 7, "ev_occupancy": 2}, {"kernel": 5}, {"slim": 1, "occupancy": 3}, {"slim": 1}, {"kernel": 3}, {"kernel": 4}, {"kernel": 2},
                                  {"kernel": 1}, {"occupancy": 1}, {"xcd": 1}, {"prefetch": 16}, {"specialise": 0}, {"moments": 0},
                                  {"stagger": 8}, {"kernel": 6, "grid": 5}, {"kernel": 6, "grid": 5, "moments": 0},
                                  {"ev_resident": 0}, {"ev_resident": 0, "kernel": 7}, {"ev_resident": 2}, {"ev_resident": 1, "grid": 7}, {"ev_resident": 1, "grid": 1}, {"ev_resident": 1, "grid": 100000}])
def test_pihna_option_sets(oracle, opts):
    """Every non-default kernel selection (rdc_set_option) of the PIHNA/TET4 path stays on the oracle."""
    conn, xyz = synth.kuhn_tet_mesh(9, order="lex")
    u = synth.pihna_fields(xyz)
    p = pihna_params_from_dict(synth.pihna_param_dict("shipped"))
    _, _, val0, rhs0 = oracle.assemble(0, 4, conn, xyz, 5, p, u_old=u)
    with AssemblyContext(0) as ctx:
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.mesh_upload(4, conn, xyz, 5)
        ctx.field_upload(FIELD_OLD_SOLUTION, u)
        ctx.assemble_pihna(p)
        val, rhs = ctx.csr_download()
    assert rel(val, val0) < TOL and rel(rhs, rhs0) < TOL


@pytest.mark.parametrize("pvariant", ["full", "realexp", "taxis_v_only"])
@pytest.mark.parametrize("opts", [{}, {"ev_occupancy": 2}, {"ev_general": 0}, {"specialise": 0}, {"part": 1}, {"ev_resident": 2}, {"ev_resident": 2, "grid": 3}])
def test_pihna_general_parameter_kernels(oracle, pvariant, opts):
    """PIHNA / TET4 with any parameter values: the element-visit kernel with all 22 moments (default, at both register budgets),
    the pair kernel it replaced ("ev_general" = 0), and the general kernel on the shipped values ("specialise" = 0)."""
    conn, xyz = synth.kuhn_tet_mesh(10, order="random")
    u = synth.pihna_fields(xyz)
    if pvariant == "taxis_v_only":
        d = synth.pihna_param_dict("shipped")
        d.update({"taxis/v": 0.3, "uptake/a/from/v": 2.0e-5})
    else:
        d = synth.pihna_param_dict("shipped" if "specialise" in opts else pvariant)
    p = pihna_params_from_dict(d)
    _, _, val0, rhs0 = oracle.assemble(0, 4, conn, xyz, 5, p, u_old=u)
    with AssemblyContext(0) as ctx:
        two_part = opts.get("part", 0)
        if two_part:
            ctx.set_option("interior_nodes", int(0.4 * xyz.shape[0]))
        for k, v in opts.items():
            if k != "part":
                ctx.set_option(k, v)
        ctx.mesh_upload(4, conn, xyz, 5)
        ctx.field_upload(FIELD_OLD_SOLUTION, u)
        ctx.timing_enable(True)
        if two_part:
            ctx.assemble_pihna_part(p, 1)
            assert ctx.part1_nodes() > 0                     # the element-visit lists serve the two parts of the general kernel too
            ctx.assemble_pihna_part(p, 2)
        else:
            ctx.assemble_pihna(p)
        val, rhs = ctx.csr_download()
    assert rel(val, val0) < TOL and rel(rhs, rhs0) < TOL


@pytest.mark.parametrize("persistent", [0, 2])
@pytest.mark.parametrize("frac", [0.0, 0.37, 1.0])
def test_two_part_assembly_equals_whole(frac, persistent):
    """rdc_set_option("part", 1|2): the rows of the leading workgroups inside [0, interior_nodes) and then the rest give
    the matrix and residual of one whole call (to round-off: the order of the LDS atomic adds is not fixed), and
    part 1 writes nothing outside its rows."""
    conn, xyz = synth.kuhn_tet_mesh(12, order="lex")
    u = synth.pihna_fields(xyz)
    p = pihna_params_from_dict(synth.pihna_param_dict("shipped"))
    n_int = int(frac * xyz.shape[0])
    with AssemblyContext(0) as ctx:
        ctx.set_option("interior_nodes", n_int)            # known before the upload: the work lists respect the split
        if persistent == 2:
            ctx.set_option("ev_resident", 2)               # k_tet4_evq on each part's piece of the permuted cluster order, a counter per part
        ctx.mesh_upload(4, conn, xyz, 5)
        ctx.field_upload(FIELD_OLD_SOLUTION, u)
        ctx.assemble_pihna(p)
        val0, rhs0 = ctx.csr_download()
        rp, _ = ctx.csr_pattern()
        ctx.field_upload(FIELD_OLD_SOLUTION, 0.5 * u)      # overwrite every row with something else
        ctx.assemble_pihna(p)
        val1, rhs1 = ctx.csr_download()
        assert not np.array_equal(val0, val1)
        ctx.field_upload(FIELD_OLD_SOLUTION, u)
        ctx.set_option("interior_nodes", n_int)
        ctx.set_option("part", 1)
        ctx.assemble_pihna(p)
        vala, rhsa = ctx.csr_download()
        ctx.set_option("part", 2)
        ctx.assemble_pihna(p)
        valb, rhsb = ctx.csr_download()
        ctx.set_option("part", 0)
    row_of = np.repeat(np.arange(rp.size - 1), np.diff(rp))
    new_rows = np.union1d(np.flatnonzero(rhsa != rhs1), np.unique(row_of[vala != val1]))
    assert new_rows.size == 0 or new_rows.max() < 5 * n_int          # part 1 stays inside the interior rows
    if frac == 1.0:
        assert rel(vala, val0) < 1e-13 and rel(rhsa, rhs0) < 1e-13
    elif frac > 0:
        assert new_rows.size > 0.5 * 5 * n_int                       # ... and covers most of them
    first_untouched = 5 * n_int
    assert np.array_equal(vala[rp[first_untouched]:], val1[rp[first_untouched]:])
    assert rel(valb, val0) < 1e-13 and rel(rhsb, rhs0) < 1e-13


@pytest.mark.parametrize("nen", [4, 8])
@pytest.mark.parametrize("model", ["pihna", "ripf", "hcc", "adpm"])
def test_shipped_pattern_instantiations_agree_with_the_general_ones(model, nen):
    """the parameter-pattern instantiations (selected from the VALUES of the four shipped input files) drop products that
    upstream multiplies by zero: with "specialise" = 0 the general instantiation runs on the same parameters and the
    results agree to rounding (advisor finding, round 1)"""
    from rdcfes_amd import FIELD_ELEM_TRACTS, adpm_params_from_dict
    conn, xyz = synth.kuhn_tet_mesh(6, order="random") if nen == 4 else synth.hex_mesh(6, jitter=0.15, order="random")
    aux = tracts = None
    if model == "pihna":
        nv, p, u, call = 5, pihna_params_from_dict(synth.pihna_param_dict("shipped")), synth.pihna_fields(xyz), "assemble_pihna"
    elif model == "ripf":
        nv, p, call = 3, ripf_params_from_dict(synth.ripf_param_dict("shipped")), "assemble_ripf"
        u, aux = synth.ripf_fields(xyz)
    elif model == "hcc":
        nv, p, u, call = 3, hcc_params_from_dict(synth.hcc_param_dict("shipped")), synth.hcc_fields(xyz), "assemble_hcc"
    else:
        nv, p, call = 3, adpm_params_from_dict(synth.adpm_param_dict("shipped"), time=3.0), "assemble_adpm"
        u, tracts = synth.adpm_fields(xyz, conn.shape[0])
    out = []
    for special in (1, 0):
        with AssemblyContext(0) as ctx:
            ctx.set_option("specialise", special)
            ctx.mesh_upload(nen, conn, xyz, nv)
            ctx.field_upload(FIELD_OLD_SOLUTION, u)
            if aux is not None:
                ctx.field_upload(FIELD_AUX_NODAL, aux)
            if tracts is not None:
                ctx.field_upload(FIELD_ELEM_TRACTS, tracts)
            getattr(ctx, call)(p)
            out.append(ctx.csr_download())
    (val1, rhs1), (val0, rhs0) = out
    assert rel(val1, val0) < 1e-13 and rel(rhs1, rhs0) < 1e-13
    # entry by entry: absolute differences at the rounding level of the largest entry of the row block
    assert np.abs(val1 - val0).max() <= 1e-12 * np.abs(val0).max()


def test_two_part_assembly_ripf_element_visits(oracle):
    """the all-terms RIPF instantiation runs on element visits (k_tet4_evc): two-part assembly on its cluster lists, and the
    pair kernel ("kernel" = 5) as the cross-check"""
    conn, xyz = synth.kuhn_tet_mesh(10, order="lex")
    u, aux = synth.ripf_fields(xyz)
    p = ripf_params_from_dict(synth.ripf_param_dict("full"))
    _, _, val0, rhs0 = oracle.assemble(1, 4, conn, xyz, 3, p, u_old=u, aux=aux)
    n_int = int(0.4 * xyz.shape[0])
    with AssemblyContext(0) as ctx:
        ctx.set_option("interior_nodes", n_int)
        ctx.mesh_upload(4, conn, xyz, 3)
        ctx.field_upload(FIELD_OLD_SOLUTION, u)
        ctx.field_upload(FIELD_AUX_NODAL, aux)
        ctx.assemble_ripf(p)
        val, rhs = ctx.csr_download()
        assert rel(val, val0) < TOL and rel(rhs, rhs0) < TOL
        rp, _ = ctx.csr_pattern()
        ctx.field_upload(FIELD_OLD_SOLUTION, 0.5 * u)
        ctx.assemble_ripf(p)
        val1, rhs1 = ctx.csr_download()
        ctx.field_upload(FIELD_OLD_SOLUTION, u)
        ctx.set_option("part", 1)
        ctx.assemble_ripf(p)
        vala, rhsa = ctx.csr_download()
        ctx.set_option("part", 2)
        ctx.assemble_ripf(p)
        valb, rhsb = ctx.csr_download()
        ctx.set_option("part", 0)
        ctx.set_option("kernel", 5)
        ctx.assemble_ripf(p)
        valc, rhsc = ctx.csr_download()
    row_of = np.repeat(np.arange(rp.size - 1), np.diff(rp))
    new_rows = np.union1d(np.flatnonzero(rhsa != rhs1), np.unique(row_of[vala != val1]))
    assert new_rows.size > 0.5 * 3 * n_int and new_rows.max() < 3 * n_int     # part 1: most interior rows, nothing else
    assert rel(valb, val0) < TOL and rel(rhsb, rhs0) < TOL
    assert rel(valc, val0) < TOL and rel(rhsc, rhs0) < TOL


def test_two_part_assembly_fallback_paths():
    """Paths that cannot launch sub-ranges (the HEX8 pair kernels, the coloured strategy) write nothing in part 1 and
    everything in part 2 -- the contract rdc_assembly.h states for rdc_set_option("part").  (The HEX8 cluster kernels do
    launch sub-ranges since round 3: tests/test_gpu_cfg5_parts.py.)"""
    conn, xyz = synth.hex_mesh(6, jitter=0.1, order="random")
    u = synth.hcc_fields(xyz)
    p = hcc_params_from_dict(synth.hcc_param_dict("full"))
    for nen, strategy, hex_kernel in [(8, SCATTER_ROWGATHER, 1), (8, SCATTER_COLOURED, 0)]:
        with AssemblyContext(0) as ctx:
            ctx.set_option("hex_kernel", hex_kernel)     # 1 = (node, element) pair kernels
            ctx.set_scatter(strategy)
            ctx.mesh_upload(nen, conn, xyz, 3)
            ctx.field_upload(FIELD_OLD_SOLUTION, u)
            ctx.assemble_hcc(p)
            val0, rhs0 = ctx.csr_download()
            ctx.field_upload(FIELD_OLD_SOLUTION, 0.5 * u)
            ctx.assemble_hcc(p)
            val1, rhs1 = ctx.csr_download()
            ctx.field_upload(FIELD_OLD_SOLUTION, u)
            ctx.set_option("interior_nodes", xyz.shape[0] // 2)
            ctx.set_option("part", 1)
            ctx.assemble_hcc(p)
            vala, rhsa = ctx.csr_download()
            assert np.array_equal(vala, val1) and np.array_equal(rhsa, rhs1)     # untouched
            assert ctx.part1_nodes() == 0
            ctx.set_option("part", 2)
            ctx.assemble_hcc(p)
            valb, rhsb = ctx.csr_download()
            assert rel(valb, val0) < 1e-13 and rel(rhsb, rhs0) < 1e-13


@pytest.mark.parametrize("moments", [1, 0])
def test_pihna_shipped_pattern_branches(oracle, moments):
    """The shipped-pattern PIHNA kernels (moment form and coefficient form) on a state that exercises the clamped
    branches on the GPU too: saturated and empty crowding, vascular fraction at 0 and 1, vasculature below the diffusion
    threshold, and a patch of all-zero nodes (0/0 in the vascular fraction -> NaN in the same entries as the oracle)."""
    conn, xyz = synth.kuhn_tet_mesh(8, order="random")
    rng = np.random.default_rng(5)
    u = synth.pihna_fields(xyz)
    p = pihna_params_from_dict(synth.pihna_param_dict("shipped"))
    x = xyz[:, 0]
    u[x < 0.25, :4] *= 60.0                                   # Te >= 1
    u[(x >= 0.25) & (x < 0.4), 3] = rng.uniform(0.0, 2.0 * p.cells_min_capacity, int(((x >= 0.25) & (x < 0.4)).sum()))
    u[(x >= 0.4) & (x < 0.5), 1:3] = 0.0                      # Ve = 1
    u[(x >= 0.5) & (x < 0.6), 3] = 0.0                        # Ve = 0
    u[x > 0.7] = 0.0                                          # empty
    _, _, val0, rhs0 = oracle.assemble(0, 4, conn, xyz, 5, p, u_old=u)
    with AssemblyContext(0) as ctx:
        ctx.set_option("moments", moments)
        ctx.mesh_upload(4, conn, xyz, 5)
        ctx.field_upload(FIELD_OLD_SOLUTION, u)
        ctx.assemble_pihna(p)
        val, rhs = ctx.csr_download()
    assert np.array_equal(np.isnan(val), np.isnan(val0)) and np.array_equal(np.isnan(rhs), np.isnan(rhs0))
    assert np.isnan(val0).any()
    ok, okr = ~np.isnan(val0), ~np.isnan(rhs0)
    assert rel(val[ok], val0[ok]) < TOL and rel(rhs[okr], rhs0[okr]) < TOL
    np.testing.assert_allclose(val[ok], val0[ok], rtol=1e-9, atol=1e-12 * np.abs(val0[ok]).max())


@pytest.mark.parametrize("resident", [0, 2])
def test_two_part_assembly_on_two_streams(oracle, resident):
    """The stream contract of rdc_set_option("part") (include/rdc_assembly.h): part 1 on one stream while the ghost rows
    of the bound solution are rewritten on another, part 2 behind that rewrite on the second stream.  Part 1 must not
    read (or publish) a ghost value, part 2 must see the new ones whatever the order in which the GPU runs the two."""
    import torch
    from rdcfes_amd import partition
    conn, xyz = synth.kuhn_tet_mesh(20, order="lex")
    u = synth.pihna_fields(xyz)
    p = pihna_params_from_dict(synth.pihna_param_dict("shipped"))
    part = partition.partition_rcb(xyz[conn].mean(axis=1), 2)
    lp = partition.build_local(conn, xyz, part, 0, 2)
    assert 0 < lp.n_interior < lp.n_owned < lp.node_global.size
    dev = torch.device("cuda", 0)
    main_s, side_s = torch.cuda.current_stream(dev), torch.cuda.Stream(device=dev)
    ghost_idx = torch.arange(lp.n_owned, lp.node_global.size, device=dev)
    lu = u[lp.node_global]
    with AssemblyContext(0) as ctx:
        ctx.set_stream(main_s.cuda_stream)
        ctx.set_option("ev_resident", resident)             # 2: both parts through k_tet4_evq (one cluster counter per part)
        ctx.mesh_upload(4, lp.conn, lp.xyz, 5, n_owned=lp.n_owned)
        u_t = torch.from_numpy(lu.copy()).to(dev)
        ctx.field_bind_device(FIELD_OLD_SOLUTION, u_t.data_ptr(), u_t.numel())
        ctx.set_option("interior_nodes", int(lp.n_interior))
        for step in range(4):
            scale = 1.0 + 0.25 * step                       # this step's ghost values
            new_ghosts = torch.from_numpy(lu[lp.n_owned:] * scale).to(dev)
            torch.cuda.synchronize()
            u_t[lp.n_owned:] = float("nan")                 # what part 1 would pick up if it looked at a ghost
            torch.cuda.synchronize()
            side_s.wait_stream(main_s)
            ctx.set_option("part", 1)
            ctx.assemble_pihna(p)                           # main stream
            with torch.cuda.stream(side_s):
                u_t.index_copy_(0, ghost_idx, new_ghosts)   # stands in for the halo exchange
            ctx.set_stream(side_s.cuda_stream)
            ctx.set_option("part", 2)
            ctx.assemble_pihna(p)                           # side stream, behind the "exchange"
            ctx.set_stream(main_s.cuda_stream)
            main_s.wait_stream(side_s)
            ctx.set_option("part", 0)
            val, rhs = ctx.csr_download()
            exp_u = lu.copy()
            exp_u[lp.n_owned:] *= scale
            _, _, val0, rhs0 = oracle.assemble(0, 4, lp.conn, lp.xyz, 5, p, u_old=exp_u, n_owned=lp.n_owned)
            assert np.isfinite(val).all() and np.isfinite(rhs).all()
            assert rel(val, val0) < TOL and rel(rhs, rhs0) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("params,opts", [("shipped", {}), ("full", {}), ("shipped", {"slim": 1}), ("shipped", {"kernel": 5}),
                                         ("shipped", {"specialise": 0})])
def test_chunked_handback_follows_the_part1_bound(params, opts):
    """rdc_part1_nodes after a part-1 call = the rows THAT call completed, whatever kernel path the parameters and options
    select (element-visit clusters, pair workgroups, or nothing): rows [0, n1) downloaded between the parts plus rows
    [n1, n_owned) after part 2 give the whole assembly (the bound used to be predicted from the
    element-visit lists alone)."""
    conn, xyz = synth.kuhn_tet_mesh(12, order="lex")
    u = synth.pihna_fields(xyz)
    p = pihna_params_from_dict(synth.pihna_param_dict(params))
    nn = xyz.shape[0]
    n_int = int(0.6 * nn)
    with AssemblyContext(0) as ctx:
        ctx.set_option("interior_nodes", n_int)
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.mesh_upload(4, conn, xyz, 5)
        ctx.field_upload(FIELD_OLD_SOLUTION, u)
        ctx.assemble_pihna(p)
        val0, rhs0 = ctx.csr_download()
        ctx.field_upload(FIELD_OLD_SOLUTION, 0.5 * u)      # every row now holds something else
        ctx.assemble_pihna(p)
        ctx.field_upload(FIELD_OLD_SOLUTION, u)
        val = np.full_like(val0, np.nan)
        rhs = np.full_like(rhs0, np.nan)
        ctx.set_option("part", 1)
        ctx.assemble_pihna(p)
        n1 = ctx.part1_nodes()
        assert 0 <= n1 <= n_int
        ctx.csr_download_rows(0, n1, val.ctypes.data, rhs.ctypes.data)
        ctx.set_option("part", 2)
        ctx.assemble_pihna(p)
        ctx.csr_download_rows(n1, nn, val.ctypes.data, rhs.ctypes.data)
        ctx.set_option("part", 0)
    assert np.isfinite(val).all() and np.isfinite(rhs).all()
    assert rel(val, val0) < 1e-13 and rel(rhs, rhs0) < 1e-13
    if not opts and params == "shipped":
        assert n1 > 0.3 * n_int                             # the default path does split


@pytest.mark.gpu
@pytest.mark.parametrize("moments", [1, 0])
def test_pihna_zero_cell_sum_with_positive_vasculature(oracle, moments):
    """c + h + v == 0 with v > 0 at the quadrature points (c = -v on a slab of nodes; not reachable after check_solution's
    clamp, but the IEEE quotient is defined): upstream's Ve_ = v / (c + h + v) is +inf and takes the Ve_ >= 1 branch
    (src/pihna.C:477-498); the device reciprocal must do the same (rcp(): v_div_fixup after the Newton steps), not NaN."""
    conn, xyz = synth.kuhn_tet_mesh(8, order="random")
    u = synth.pihna_fields(xyz)
    p = pihna_params_from_dict(synth.pihna_param_dict("shipped"))
    slab = (xyz[:, 0] > 0.3) & (xyz[:, 0] < 0.8)
    u[slab, 3] = 7170.0
    u[slab, 1] = -7170.0
    u[slab, 2] = 0.0
    inside = slab[conn].all(axis=1)
    assert inside.sum() > 100                                   # elements whose every point has c + h + v == 0 exactly
    _, _, val0, rhs0 = oracle.assemble(0, 4, conn, xyz, 5, p, u_old=u)
    with AssemblyContext(0) as ctx:
        ctx.set_option("moments", moments)
        ctx.mesh_upload(4, conn, xyz, 5)
        ctx.field_upload(FIELD_OLD_SOLUTION, u)
        ctx.assemble_pihna(p)
        val, rhs = ctx.csr_download()
    assert np.array_equal(np.isnan(val), np.isnan(val0)) and np.array_equal(np.isnan(rhs), np.isnan(rhs0))
    ok, okr = ~np.isnan(val0), ~np.isnan(rhs0)
    assert ok.mean() > 0.5
    assert rel(val[ok], val0[ok]) < TOL and rel(rhs[okr], rhs0[okr]) < TOL


@pytest.mark.gpu
def test_part1_rows_travel_while_part2_runs():
    """rdc_csr_download_rows_async right behind part 1: the copy starts behind the work enqueued so far and is independent of
    part 2, which is enqueued next on the context's stream; the ticket says when the rows are in host memory."""
    import ctypes as C
    from rdcfes_amd import _lib
    L = _lib.load()
    conn, xyz = synth.kuhn_tet_mesh(14, order="lex")
    u = synth.pihna_fields(xyz)
    p = pihna_params_from_dict(synth.pihna_param_dict("shipped"))
    nn = xyz.shape[0]
    with AssemblyContext(0) as ctx:
        ctx.set_option("interior_nodes", int(0.5 * nn))
        ctx.mesh_upload(4, conn, xyz, 5)
        ctx.field_upload(FIELD_OLD_SOLUTION, u)
        ctx.assemble_pihna(p)
        val0, rhs0 = ctx.csr_download()
        rp, _ = ctx.csr_pattern()
        val = np.full_like(val0, np.nan)
        rhs = np.full_like(rhs0, np.nan)
        assert L.rdc_host_pin(ctx._h, val.ctypes.data, val.nbytes) == 0 and L.rdc_host_pin(ctx._h, rhs.ctypes.data, rhs.nbytes) == 0
        ctx.field_upload(FIELD_OLD_SOLUTION, 0.5 * u)
        ctx.assemble_pihna(p)
        ctx.field_upload(FIELD_OLD_SOLUTION, u)
        ctx.assemble_pihna_part(p, 1)
        n1 = ctx.part1_nodes()
        t1, t2 = C.c_int(-1), C.c_int(-1)
        assert L.rdc_csr_download_rows_async(ctx._h, 0, n1, val.ctypes.data, rhs.ctypes.data, C.byref(t1)) == 0
        ctx.assemble_pihna_part(p, 2)                          # enqueued behind the copy's fence, not waited for by it
        assert L.rdc_ticket_wait(ctx._h, t1.value) == 0
        assert np.isfinite(val[:rp[5 * n1]]).all() and np.isnan(val[rp[5 * n1]:]).all()     # exactly the rows of part 1 have landed
        assert L.rdc_csr_download_rows_async(ctx._h, n1, nn, val.ctypes.data, rhs.ctypes.data, C.byref(t2)) == 0
        assert L.rdc_ticket_wait(ctx._h, t2.value) == 0
        assert L.rdc_ticket_wait(ctx._h, 15) != 0               # a ticket that was never issued
        assert L.rdc_host_unpin(ctx._h, val.ctypes.data) == 0 and L.rdc_host_unpin(ctx._h, rhs.ctypes.data) == 0
    assert n1 > 0
    assert rel(val, val0) < 1e-13 and rel(rhs, rhs0) < 1e-13
