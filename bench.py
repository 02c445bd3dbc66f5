#!/usr/bin/env python3
"""bench.py — elements assembled per second of the PIHNA element-assembly hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one invocation of the assemble callback over the whole mesh: (N > 1: halo exchange of
ghost-node DoFs over RCCL, then) evaluation of Ke/Fe for every element and scatter into the global
CSR matrix + rhs.  Inputs are resident in HBM when the timed region starts.

Workload: BASELINE.json's metric is quoted on the 10M-tet PIHNA mesh, which fits one GPU:
K(119) = 10,110,954 TET4 / 1,728,000 nodes, 5 unknowns (8.64 M DoFs, 648 M CSR values), synthetic
fields, parameters of run/PIHNA/input.dat.  N > 1 partitions that SAME mesh (strong scaling).
Prints ONE JSON line on rank 0.  At N = 1 the line also carries
  "configs"      kernel time / elements per second / roofline fraction of the other BASELINE configurations
                 (cfg2 PIHNA K(55), cfg3 RIPF K(94), cfg5 HCC H(126) + solid H(126)) and of the general-parameter PIHNA
                 kernel on K(119), each timed here with HIP events (3 warm-ups + 10 launches);
  "handback"     the CSR hand-back to the host (SURVEY §8d "kernel + CSR-handback separately"): D2H of values + rhs
                 into pinned memory, alone, behind the kernel, and overlapped with part 2 of a two-part assembly;
  "cpu_baseline" the oracle (a port of the reference loop) on the benchmark mesh itself with all host cores (+ 1 core on a sample);
  "parity"       relative residual of the timed configuration's GPU result (rhs L2, matrix Frobenius) against that CPU assembly;
  "value_with_handback"  elements per second when the CSR is downloaded to the host every step (never `value`).
At N > 1 it carries "multi_gpu": halo bytes, ghost-element overhead, halo time, host time per step.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
FP64_PEAK_TFLOPS = 78.6  # vector FP64: 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz


def algorithmic_bytes(nen, n_elem, n_node, n_owned, nvar, n_in, nnz, solid=False):
    """SURVEY §8d / BASELINE.md §3: compulsory traffic of one assembly pass (FP64 values, int32 ids).
    solid: current + undeformed coordinates, 3 fibre + 1 subdomain entries per element, no nodal input fields."""
    b = 4 * nen * n_elem + 8 * 3 * n_node + 8 * n_in * n_node + 8 * nnz + 8 * nvar * n_owned
    if solid:
        b += 8 * 3 * n_node + 8 * 4 * n_elem
    return b


def host_threads():
    """cores this process may use, capped at the GPU box's share per GPU (16)"""
    return max(1, min(16, len(os.sched_getaffinity(0))))


def cpu_baseline(conn, xyz, u, p, mesh_name, n_sample, param_variant, gpu_val=None, gpu_rhs=None):
    """Oracle ("port" of the reference loop + MatSetValues-like insertion), timing build (-O3 -march=native, compiled
    here) ON THE BENCHMARK MESH ITSELF with all host cores (rows split over threads: the stand-in for `mpiexec -n P`),
    and on one core on a K(n_sample) sample (config 1 is "1 MPI rank").  With the GPU result of the timed configuration
    given, also the parity residual of that result against this same-mesh CPU assembly ("same mesh, same
    fields, same run ... report elem/s, speed-up, and parity residual").  -> (cpu_baseline, parity)"""
    from oracle import oracle as O
    from rdcfes_amd import pihna_params_from_dict, synth
    O.fast_lib()
    nt = host_threads()
    t0 = time.perf_counter()
    pattern = O.build_pattern(4, conn, xyz.shape[0], xyz.shape[0], 5)[:2]
    t_pat = time.perf_counter() - t0
    t0 = time.perf_counter()
    _, _, val0, rhs0 = O.assemble(O.MODEL_PIHNA, 4, conn, xyz, 5, p, u_old=u, pattern=pattern, threads=nt, fast=True)
    dt_all = time.perf_counter() - t0
    parity = None
    if gpu_val is not None:
        def rel(a, b):   # chunked: no 5 GB temporaries
            num = den = 0.0
            for i in range(0, a.size, 1 << 24):
                d = a[i:i + (1 << 24)] - b[i:i + (1 << 24)]
                num += float(np.dot(d, d))
                den += float(np.dot(b[i:i + (1 << 24)], b[i:i + (1 << 24)]))
            return (num / max(den, 1e-300)) ** 0.5
        parity = {"rhs_rel_l2": rel(gpu_rhs, rhs0), "matrix_rel_fro": rel(gpu_val, val0), "tolerance": 1e-10,
                  "against": f"oracle (CPU port) on the benchmark mesh {mesh_name}, same fields and parameters, whole CSR matrix "
                             f"({val0.size} values) and rhs ({rhs0.size} entries) of the timed configuration's last step"}
    n_all = conn.shape[0]
    del val0, rhs0, pattern
    out = {"value": n_all / dt_all, "unit": "elements/s", "cores": nt, "kind": "port",
           "sample": f"the benchmark mesh itself, {mesh_name} = {n_all} TET4, same fields / parameters as the GPU run, full assembly incl. "
                     f"sorted-row CSR insertion: {dt_all:.1f} s on {nt} cores (rows split over OpenMP threads, elements straddling "
                     f"two ranges evaluated twice; pattern built beforehand in {t_pat:.1f} s, not counted, as for the GPU) "
                     "(oracle/rdc_oracle.c, gcc -O3 -march=native); stand-in for the reference's libMesh/PETSc path, which "
                     "cannot be built here"}
    if n_sample > 0:
        conn1, xyz1 = synth.kuhn_tet_mesh(n_sample, order="lex")
        u1 = synth.pihna_fields(xyz1)
        p1 = pihna_params_from_dict(synth.pihna_param_dict(param_variant))
        pat1 = O.build_pattern(4, conn1, xyz1.shape[0], xyz1.shape[0], 5)[:2]
        t0 = time.perf_counter()
        O.assemble(O.MODEL_PIHNA, 4, conn1, xyz1, 5, p1, u_old=u1, pattern=pat1, threads=1, fast=True)
        dt_1 = time.perf_counter() - t0
        out["value_1core"] = conn1.shape[0] / dt_1
        out["sample_1core"] = f"K({n_sample}) = {conn1.shape[0]} TET4 of the same generator: {dt_1:.1f} s on 1 core"
    return out, parity


def time_config(name, nen, conn, xyz, nvar, setup, call, n_in, solid=False, reps=10, warm=3, note=None):
    """kernel time of one configuration with HIP events on the context stream -> dict for the "configs" array"""
    from rdcfes_amd import AssemblyContext
    with AssemblyContext(0) as ctx:
        t0 = time.perf_counter()
        ctx.mesh_upload(nen, conn, xyz, nvar)
        prep_s = time.perf_counter() - t0
        setup(ctx)
        for _ in range(warm):
            call(ctx)
        ctx.synchronize()
        ctx.timing_enable(True)
        for _ in range(reps):
            call(ctx)
        samples = ctx.timing_samples_ms()
        ctx.timing_enable(False)
        n = len(samples)
        ms = float(sum(samples)) / max(n, 1)
        _, nnz = ctx.csr_dims()
    b = algorithmic_bytes(nen, conn.shape[0], xyz.shape[0], xyz.shape[0], nvar, n_in, nnz, solid)
    out = {"workload": name, "elements": int(conn.shape[0]), "nodes": int(xyz.shape[0]), "nnz": int(nnz),
           "kernel_ms": ms, "elements_per_s": conn.shape[0] / (ms * 1e-3), "algorithmic_bytes_per_launch": int(b),
           "achieved_GBps": b / (ms * 1e-3) / 1e9, "frac": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "launches_timed": int(n),
           "kernel_ms_median": float(np.median(samples)) if samples else None, "kernel_ms_min": float(min(samples)) if samples else None,
           "host_prep_s": round(prep_s, 2)}
    if note:
        out["note"] = note
    # HBM traffic and FP64 rate of this configuration's kernels from the committed counter passes (profiles/pmc_configs.json, tools/make_profiles_configs.sh),
    # only when they were taken from the kernel sources built now
    try:
        from rdcfes_amd import build as B
        t = json.loads((ROOT / "profiles" / "pmc_configs.json").read_text())
        if t.get("source_hash") == B.source_hash():
            for key, v in t["kernels"].items():
                if name.startswith(key):
                    out["traffic"] = int(v["hbm_bytes_per_launch"])
                    out["fp64_tflops"] = v["fp64_flop_per_launch"] / (ms * 1e-3) / 1e12
                    out["fp64_frac"] = out["fp64_tflops"] / FP64_PEAK_TFLOPS
                    out["profile_kernel"] = v["kernel"]
    except Exception:
        pass
    return out


def extra_configs():
    """the BASELINE configurations beside the headline one, each at its stated size (SURVEY §8d synthetic inputs)"""
    from rdcfes_amd import (SolidMaterial, SolidParams, hcc_params_from_dict, pihna_params_from_dict, ripf_params_from_dict, synth)
    from rdcfes_amd.context import FIELD_AUX_NODAL, FIELD_ELEM_FIBRE, FIELD_OLD_SOLUTION, FIELD_UNDEFORMED_XYZ
    out = []
    # cfg2: PIHNA, 1M TET4
    conn, xyz = synth.kuhn_tet_mesh(55, order="lex")
    u = synth.pihna_fields(xyz)
    p = pihna_params_from_dict(synth.pihna_param_dict("shipped"))
    out.append(time_config("cfg2: PIHNA TET4 K(55), params run/PIHNA/input.dat", 4, conn, xyz, 5,
                           lambda c: c.field_upload(FIELD_OLD_SOLUTION, u), lambda c: c.assemble_pihna(p), 5))
    # general-parameter PIHNA kernel on the metric's mesh (every transport term on: no parameter-pattern variant applies)
    conn, xyz = synth.kuhn_tet_mesh(119, order="lex")
    u = synth.pihna_fields(xyz)
    pf = pihna_params_from_dict(synth.pihna_param_dict("full"))
    out.append(time_config("PIHNA TET4 K(119), all transport terms non-zero (general instantiation)", 4, conn, xyz, 5,
                           lambda c: c.field_upload(FIELD_OLD_SOLUTION, u), lambda c: c.assemble_pihna(pf), 5))
    # cfg3: RIPF, 5M TET4
    conn, xyz = synth.kuhn_tet_mesh(94, order="lex")
    u, aux = synth.ripf_fields(xyz)

    def setup_ripf(c):
        c.field_upload(FIELD_OLD_SOLUTION, u)
        c.field_upload(FIELD_AUX_NODAL, aux)
    for pv in ("shipped", "full"):
        pr = ripf_params_from_dict(synth.ripf_param_dict(pv))
        out.append(time_config(f"cfg3: RIPF TET4 K(94), params run/RIPF133/input.dat ({pv})", 4, conn, xyz, 3, setup_ripf,
                               lambda c: c.assemble_ripf(pr), 6))
    # cfg5: coupled HCC + solid, 2M HEX8 (one reaction-diffusion assembly on the deformed mesh, one Newton assembly)
    conn, Xu = synth.hex_mesh(126, jitter=0.1, order="lex")
    x = Xu + synth.solid_displacement(Xu, amp=0.02 / 126 * 8)
    uh = synth.hcc_fields(Xu)
    ph = hcc_params_from_dict(synth.hcc_param_dict("full"))
    out.append(time_config("cfg5 (RD half): HCC HEX8 H(126) on the deformed mesh, all rates non-zero", 8, conn, x, 3,
                           lambda c: c.field_upload(FIELD_OLD_SOLUTION, uh), lambda c: c.assemble_hcc(ph), 3, reps=6))
    ps = hcc_params_from_dict(synth.hcc_param_dict("shipped"))
    out.append(time_config("cfg5 (RD half, shipped): HCC HEX8 H(126) on the deformed mesh, params run/Coupled/HCC/input.dat (every rate zero)", 8, conn, x, 3,
                           lambda c: c.field_upload(FIELD_OLD_SOLUTION, uh), lambda c: c.assemble_hcc(ps), 3, reps=6))
    em = (np.linalg.norm(Xu[conn].mean(axis=1) - 0.5, axis=1) < 0.3).astype(np.int32)
    mats = [SolidMaterial(2.0e3, 0.4, 0.0, (0.0, 0.0, 0.0)), SolidMaterial(2.0e3, 0.4, 0.0, (0.3, 0.3, 0.3))]
    se0, ss0 = synth.boundary_sides(8, conn, Xu, 2, 0.0)
    sd = np.zeros((se0.size, 3))
    sp = SolidParams(0.4, 1.0e8, 0, 0)
    fibre = np.tile([0.0, 0.0, 1.0], (conn.shape[0], 1))

    def setup_solid(c):
        c.field_upload(FIELD_UNDEFORMED_XYZ, Xu)
        c.field_upload(FIELD_ELEM_FIBRE, fibre)
        c.solid_set_materials(em, mats)
        c.solid_set_sides(se0, ss0, sd)
    out.append(time_config("cfg5 (solid half): SolidSystem HEX8 H(126), residual + Jacobian of one Newton iteration", 8, conn, x, 3,
                           setup_solid, lambda c: c.solid_assemble(sp, True), 0, solid=True, reps=6,
                           note="FP64-compute-bound path (SURVEY §8d): frac is against the HBM roof for completeness"))
    return out


def handback(ctx, assemble, n_owned, nnz, n_rows, kern_ms, reps=3):
    """CSR hand-back to the host (the reference hands its blocks to PETSc, src/pihna.C:754-755): D2H of values + rhs."""
    import torch
    val_h = torch.empty(nnz, dtype=torch.float64, pin_memory=True)
    rhs_h = torch.empty(n_rows, dtype=torch.float64, pin_memory=True)
    main_s = torch.cuda.current_stream()
    copy_s = torch.cuda.Stream()
    ctx.set_option("part", 0)

    def timed(fn):
        torch.cuda.synchronize()
        fn()  # warm
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    def copy_only():
        ctx.csr_download_rows(0, n_owned, val_h.data_ptr(), rhs_h.data_ptr(), asynchronous=True)

    def serial():
        assemble()
        ctx.csr_download_rows(0, n_owned, val_h.data_ptr(), rhs_h.data_ptr(), asynchronous=True)

    # overlapped: rows of the first half of the nodes are copied while the second half is being assembled
    ctx.set_option("interior_nodes", n_owned // 2)
    n1 = ctx.part1_nodes()

    def overlapped():
        ctx.set_option("part", 1)
        assemble()
        copy_s.wait_stream(main_s)
        ctx.set_stream(copy_s.cuda_stream)
        ctx.csr_download_rows(0, n1, val_h.data_ptr(), rhs_h.data_ptr(), asynchronous=True)
        ctx.set_stream(main_s.cuda_stream)
        ctx.set_option("part", 2)
        assemble()
        ctx.set_option("part", 0)
        main_s.wait_stream(copy_s)   # one copy engine direction: keep the two transfers in order
        ctx.csr_download_rows(n1, n_owned, val_h.data_ptr(), rhs_h.data_ptr(), asynchronous=True)

    d2h = timed(copy_only)
    ser = timed(serial)
    ovl = timed(overlapped) if n1 > 0 else None
    ctx.set_option("part", 0)
    ctx.set_option("interior_nodes", -1)
    nbytes = 8 * (nnz + n_rows)
    return {"bytes": int(nbytes), "d2h_ms": d2h, "d2h_GBps": nbytes / d2h / 1e6, "kernel_ms": kern_ms,
            "kernel_plus_handback_ms": ser, "kernel_plus_handback_overlapped_ms": ovl,
            "note": "host wall time per step, pinned host buffers, rdc_csr_download_rows; the device-pointer hand-off "
                    "(rdc_csr_values_device_ptr) avoids this copy altogether"}


def state_sensitivity(ctx, p, xyz, u_t, b_alg, reps=10):
    """The element-visit kernel takes a short cut for waves all of whose elements are in the BACKGROUND state of the reference's
    shipped field file (n = c = h = a = 0, v > 0: the moments that are sums of exact zeros are not evaluated).  The headline state
    (tumour inside a sphere r = 0.25, background elsewhere) has 93 % such elements -- the shipped file 99.9 % --, so the
    kernel time depends on the state.  Timed here on the same mesh: (a) the same state with the short cut off, (b) a DENSE state
    (tumour values at every node: no background element), short cut on and off.  Kernel times by HIP events, as the headline."""
    import torch
    from rdcfes_amd import synth

    def run(label):
        for _ in range(3):
            ctx.assemble_pihna(p)
        ctx.synchronize()
        ctx.timing_enable(True)
        for _ in range(reps):
            ctx.assemble_pihna(p)
        ms, k = ctx.timing_sum_ms()
        ctx.timing_enable(False)
        return {"state": label, "kernel_ms": ms / k, "roofline_frac": b_alg / (ms / k * 1e-3) / 1e9 / HBM_PEAK_GBPS}

    out = []
    keep = u_t.clone()
    bg = ((keep[:, [0, 1, 2, 4]] == 0).all(dim=1) & (keep[:, 3] > 0)).double().mean().item()
    ctx.set_option("ev_background", 0)
    out.append(run(f"headline state ({bg:.1%} of the nodes in the background state), short cut OFF"))
    dense = torch.from_numpy(np.ascontiguousarray(synth.pihna_fields(xyz, radius=10.0))).to(u_t.device)
    u_t.copy_(dense)
    out.append(run("dense state (tumour values at every node), short cut OFF"))
    ctx.set_option("ev_background", 1)
    out.append(run("dense state (tumour values at every node), short cut ON (its test costs time, nothing is skipped)"))
    u_t.copy_(keep)
    ctx.synchronize()
    return out


def two_part_host_cost(ctx, p, n_owned, steps=50):
    """What the two-part step of the N > 1 path costs the HOST at N = 1 (no communication): two C-ABI calls per step
    (rdc_assemble_pihna_part on the main and on a side stream, the stream joins in between as in the N > 1 step) against
    the one call of the whole assembly, and the same two parts replayed from a hipGraph (the exchange stays outside it)."""
    import torch
    main_s, side_s = torch.cuda.current_stream(), torch.cuda.Stream()
    ctx.set_option("interior_nodes", int(0.9 * n_owned))   # no ghosts at N = 1: any prefix of the nodes is "interior"

    def two_part():
        side_s.wait_stream(main_s)
        ctx.assemble_pihna_part(p, 1, main_s.cuda_stream)
        ctx.assemble_pihna_part(p, 2, side_s.cuda_stream)
        main_s.wait_stream(side_s)

    def whole():
        ctx.assemble_pihna(p)

    def run(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        return t_host / steps * 1e6, (time.perf_counter() - t0) / steps * 1e3
    out = {}
    out["whole_call_host_us"], out["whole_call_ms_per_step"] = run(whole)
    out["two_part_host_us"], out["two_part_ms_per_step"] = run(two_part)
    try:   # both parts on one captured stream: a step is then one hipGraphLaunch
        g = torch.cuda.CUDAGraph()
        cap_s = torch.cuda.Stream()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=cap_s):
            ctx.assemble_pihna_part(p, 1, cap_s.cuda_stream)
            ctx.assemble_pihna_part(p, 2, cap_s.cuda_stream)
        out["graph_host_us"], out["graph_ms_per_step"] = run(g.replay)
    except Exception as e:   # capture is best effort: report why it failed rather than hide it
        out["graph_error"] = str(e)[:200]
        torch.cuda.synchronize()
    ctx.set_option("interior_nodes", -1)
    ctx.set_option("part", 0)
    out["note"] = ("host time to ENQUEUE one step (no device wait), and wall time per step with the device kept busy; two_part = "
                   "rdc_assemble_pihna_part x 2 on two streams with the joins of the N > 1 step, no exchange")
    return out


def profile_numbers(n, world):
    """HBM bytes and FP64 flops per launch from the committed rocprofv3 --pmc passes -- only if they were taken from
    the kernel sources that are built now (profiles/pmc_traffic.json carries their hash), else null."""
    from rdcfes_amd import build as B
    pmc = ROOT / "profiles" / "pmc_traffic.json"
    try:
        t = json.loads(pmc.read_text())
        if t.get("workload") == f"K({n})" and t.get("n_gpus") == world and t.get("source_hash") == B.source_hash():
            return t.get("hbm_bytes_per_launch"), t.get("fp64_flop_per_launch"), t.get("kernel")
    except Exception:
        pass
    return None, None, None


def run_cfg5(a):
    """BASELINE config 5 across GPUs (`--workload cfg5`): coupled_hcc + solid_system on the deforming H(n) HEX8 mesh, RCB element
    partition + one ghost layer; one step = ONE grouped halo exchange carrying the HCC unknowns and the current coordinates of
    the moved mesh (6 doubles per interface node), the HCC assembly on the moved mesh and one Newton assembly (residual +
    tangent) of the solid system -- both in two parts, the interior clusters overlapping the exchange
    (src/coupled_hcc.C:98-130 call order, src/solid_system.C:103-123 mesh move).  Prints one JSON line on rank 0."""
    import torch
    import torch.distributed as dist
    from rdcfes_amd import AssemblyContext, SolidMaterial, SolidParams, hcc_params_from_dict, partition, synth
    from rdcfes_amd.context import FIELD_ELEM_FIBRE, FIELD_OLD_SOLUTION, FIELD_UNDEFORMED_XYZ
    from rdcfes_amd.halo import HaloExchange
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the assembly path is HIP-only (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    n = a.n if a.n != 119 else 126
    conn, Xu = synth.hex_mesh(n, jitter=0.1, order="lex")
    x = Xu + synth.solid_displacement(Xu, amp=0.02 / n * 8)
    uh = synth.hcc_fields(Xu)
    ph = hcc_params_from_dict(synth.hcc_param_dict("full"))
    em = (np.linalg.norm(Xu[conn].mean(axis=1) - 0.5, axis=1) < 0.3).astype(np.int32)
    mats = [SolidMaterial(2.0e3, 0.4, 0.0, (0.0, 0.0, 0.0)), SolidMaterial(2.0e3, 0.4, 0.0, (0.3, 0.3, 0.3))]
    se, ss = synth.boundary_sides(8, conn, Xu, 2, 0.0)
    sp = SolidParams(0.4, 1.0e8, 0, 0)
    n_elem_global, n_node_global = conn.shape[0], Xu.shape[0]
    part = partition.partition_rcb(Xu[conn].mean(axis=1), world)
    lp = partition.build_local(conn, Xu, part, rank, world)
    ng, eg = lp.node_global, lp.elem_global
    g2l_e = np.full(n_elem_global, -1, dtype=np.int64)
    g2l_e[eg] = np.arange(eg.size)
    keep = g2l_e[se] >= 0                                            # boundary sides of the local elements
    l_se, l_ss = g2l_e[se[keep]], ss[keep]
    main_s, halo_s = torch.cuda.current_stream(), torch.cuda.Stream(device=dev)
    hcc, sol = AssemblyContext(local_rank), AssemblyContext(local_rank)
    for c in (hcc, sol):
        c.set_stream(main_s.cuda_stream)
        c.set_option("interior_nodes", int(lp.n_interior) if a.overlap else -1)
        c.mesh_upload(8, lp.conn, x[ng], 3, n_owned=lp.n_owned)
    u_t = torch.from_numpy(np.ascontiguousarray(uh[ng])).to(dev)
    hcc.field_bind_device(FIELD_OLD_SOLUTION, u_t.data_ptr(), u_t.numel())
    sol.field_upload(FIELD_UNDEFORMED_XYZ, Xu[ng])
    sol.field_upload(FIELD_ELEM_FIBRE, np.tile([0.0, 0.0, 1.0], (eg.size, 1)))
    sol.solid_set_materials(em[eg], mats)
    sol.solid_set_sides(l_se, l_ss, np.zeros((l_se.size, 3)))
    x_sol, x_hcc = sol.coords_tensor(), hcc.coords_tensor()
    hx = HaloExchange(lp, 6, dev) if world > 1 else None
    overlap = bool(a.overlap) and lp.n_interior > 0
    del conn, Xu, x, uh, em

    def step():
        if overlap:
            halo_s.wait_stream(main_s)
            hcc.assemble_hcc_part(ph, 1, main_s.cuda_stream)             # interior clusters: no ghost value, no ghost coordinate
            sol.solid_assemble_part(sp, True, 1, main_s.cuda_stream)
            with torch.cuda.stream(halo_s):
                if hx is not None:
                    hx.exchange_many([u_t, x_sol])                       # one message per peer: HCC unknowns + current coordinates
                x_hcc[lp.n_owned:].copy_(x_sol[lp.n_owned:])             # the HCC system runs on the same moved mesh
            hcc.assemble_hcc_part(ph, 2, halo_s.cuda_stream)
            sol.solid_assemble_part(sp, True, 2, halo_s.cuda_stream)
            main_s.wait_stream(halo_s)
            return
        if hx is not None:
            hx.exchange_many([u_t, x_sol])
        x_hcc[lp.n_owned:].copy_(x_sol[lp.n_owned:])
        hcc.assemble_hcc(ph)
        sol.solid_assemble(sp, True)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(a.warmup):
        step()
    fence()
    for c in (hcc, sol):
        c.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    t_host = time.perf_counter() - t0
    fence()
    dt = time.perf_counter() - t0
    k_hcc, k_sol = sum(hcc.timing_samples_ms()) / a.steps, sum(sol.timing_samples_ms()) / a.steps
    stats = torch.tensor([dt, k_hcc, k_sol, t_host / a.steps * 1e6, float(hx.bytes_per_step if hx else 0), float(lp.conn.shape[0]) / max(lp.n_elem_owned, 1),
                          float(lp.n_interior) / max(lp.n_owned, 1), float(hcc.part1_nodes()) / max(lp.n_owned, 1)], dtype=torch.float64,
                         device="cpu" if (world > 1 and a.backend != "nccl") else dev)
    smin = stats.clone()
    if world > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
        dist.all_reduce(smin, op=dist.ReduceOp.MIN)
    if rank == 0:
        dt = float(stats[0])
        _, nnz_h = hcc.csr_dims()
        out = {"metric": "elements_assembled_per_sec", "value": n_elem_global * a.steps / dt, "unit": "elements/s", "n_gpus": world,
               "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"cfg5: coupled HCC + SolidSystem on the deforming HEX8 mesh H({n}): {n_elem_global} hexes, {n_node_global} nodes; "
                                      "one step = one grouped halo (HCC unknowns + current coordinates) + HCC assembly (all rates on) + one Newton "
                                      "assembly (residual + tangent, penalty sides); an element counts once per step",
                          "parallelism": (f"element partition x{world} (RCB), 1 ghost layer, halo p2p over " + ("RCCL" if a.backend == "nccl" else "gloo (host-staged rehearsal)") +
                                          (", overlapped with the interior clusters" if overlap else "")) if world > 1 else "single GPU" + (", two-part path forced" if overlap else ""),
                          "rank0_local_elements": int(lp.conn.shape[0]), "rank0_nnz_per_system": int(nnz_h)},
               "multi_gpu": {"kernel_ms_hcc_max": float(stats[1]), "kernel_ms_solid_max": float(stats[2]), "kernel_ms_hcc_min": float(smin[1]),
                             "kernel_ms_solid_min": float(smin[2]), "host_enqueue_us_per_step_max": float(stats[3]),
                             "halo_send_bytes_per_rank_max": float(stats[4]), "local_over_owned_elements_max": float(stats[5]),
                             "interior_node_fraction_min": float(smin[6]), "part1_row_fraction_min": float(smin[7]),
                             "note": "kernel_ms = HIP events around the assembly kernels of a step (both parts), per system"}}
        print(json.dumps(out), flush=True)
    hcc.close()
    sol.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--workload", default="pihna", choices=["pihna", "cfg5"], help="pihna = the metric's configuration (default); cfg5 = coupled "
                    "HCC + solid on H(126) HEX8, partitioned like the PIHNA run (BASELINE config 5)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mesh-n", dest="n", type=int, default=119, help="K(n) mesh: n^3 cells x 6 tets (119 -> 10.1M tets)")
    ap.add_argument("--params", default="shipped", choices=["shipped", "full", "realexp"])
    ap.add_argument("--scatter", default="auto", choices=["auto", "coloured", "rowgather"])
    ap.add_argument("--variant", default="auto", choices=["auto", "generic"])
    ap.add_argument("--order", default="lex", choices=["lex", "random"])
    ap.add_argument("--overlap", type=int, default=1, help="N > 1: 1 = halo exchange overlapped with the assembly of interior rows, 0 = exchange first")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (production); gloo = host-staged halo, for rehearsing N > 1 on a 1-GPU box")
    ap.add_argument("--opt", action="append", default=[], help="tuning knob key=value (rdc_set_option)")
    ap.add_argument("--cpu-sample", type=int, default=60, help="K(m) sample for the 1-core CPU baseline (0 = skip it)")
    ap.add_argument("--cpu-baseline", type=int, default=1, help="N = 1: time the oracle on the benchmark mesh with all host cores and "
                    "report the parity residual of the GPU result against it (0 = skip)")
    ap.add_argument("--state-check", type=int, default=1, help="N = 1: also time the kernel on a dense state and with the background short cut off")
    ap.add_argument("--configs", type=int, default=1, help="N = 1: also time the other BASELINE configurations (0 = skip)")
    ap.add_argument("--configs-only", type=int, default=0, help="1: run ONLY the other BASELINE configurations and print their array (the command "
                    "tools/make_profiles_configs.sh profiles)")
    ap.add_argument("--two-part", dest="two_part", type=int, default=1, help="N = 1: also measure the host cost of the two-part step of the N > 1 path (0 = skip)")
    ap.add_argument("--handback", type=int, default=1, help="N = 1: also time the CSR hand-back to the host (0 = skip)")
    a = ap.parse_args()

    if a.configs_only:
        print(json.dumps({"configs": extra_configs()}), flush=True)
        return
    if a.workload == "cfg5":
        return run_cfg5(a)
    import torch
    import torch.distributed as dist
    from rdcfes_amd import AssemblyContext, partition, pihna_params_from_dict, synth
    from rdcfes_amd.context import FIELD_OLD_SOLUTION
    from rdcfes_amd.halo import HaloExchange

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the assembly path is HIP-only (no CPU fallback)")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    conn, xyz = synth.kuhn_tet_mesh(a.n, order=a.order)
    u = synth.pihna_fields(xyz)
    p = pihna_params_from_dict(synth.pihna_param_dict(a.params))
    n_elem_global, n_node_global = conn.shape[0], xyz.shape[0]
    if world > 1:
        part = partition.partition_rcb(xyz[conn].mean(axis=1), world)
        lp = partition.build_local(conn, xyz, part, rank, world)
        l_conn, l_xyz, n_owned, l_u = lp.conn, lp.xyz, lp.n_owned, u[lp.node_global]
    else:
        lp, l_conn, l_xyz, n_owned, l_u = None, conn, xyz, n_node_global, u
    del conn, xyz, u

    ctx = AssemblyContext(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)  # same stream as the halo's torch ops
    if lp is not None and a.overlap and lp.n_interior > 0:
        ctx.set_option("interior_nodes", int(lp.n_interior))   # before the upload: the work lists respect the split
    ctx.mesh_upload(4, l_conn, l_xyz, 5, n_owned=n_owned)
    ctx.set_scatter({"auto": 0, "coloured": 1, "rowgather": 2}[a.scatter])
    ctx.set_kernel_variant({"auto": 0, "generic": 1}[a.variant])
    for kv in a.opt:
        k_, v_ = kv.split("=")
        ctx.set_option(k_, int(v_))
    u_t = torch.from_numpy(np.ascontiguousarray(l_u)).to(dev)
    ctx.field_bind_device(FIELD_OLD_SOLUTION, u_t.data_ptr(), u_t.numel())
    hx = HaloExchange(lp, 5, dev) if world > 1 else None
    n_rows, nnz = ctx.csr_dims()

    # N > 1: the halo exchange runs on a side stream while the rows of interior nodes (no ghost node in any of their
    # elements; partition.build_local numbers them first) are assembled on the main stream; the remaining rows follow
    # the exchange on its stream (tools/two_part_ab.py: the split itself costs ~13 us per step at per-GPU size).
    # Stream contract of the two parts: include/rdc_assembly.h ("part").
    overlap = hx is not None and a.overlap and lp.n_interior > 0
    if overlap:
        main_s, halo_s = torch.cuda.current_stream(), torch.cuda.Stream(device=dev)
        ctx.set_option("interior_nodes", int(lp.n_interior))

    def step():
        if overlap:
            halo_s.wait_stream(main_s)      # the previous step has read the ghost rows this exchange overwrites
            ctx.assemble_pihna_part(p, 1, main_s.cuda_stream)   # interior rows, main stream, concurrent with the exchange
            with torch.cuda.stream(halo_s):
                hx.exchange(u_t)
            ctx.assemble_pihna_part(p, 2, halo_s.cuda_stream)   # rows next to ghosts, behind the exchange on its stream (fills part 1's tail)
            main_s.wait_stream(halo_s)
            return
        if hx is not None:
            hx.exchange(u_t)
        ctx.assemble_pihna(p)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    ctx.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    t_host = time.perf_counter() - t0       # host time to ENQUEUE the steps (no device wait inside step())
    fence()
    dt = time.perf_counter() - t0
    samples = ctx.timing_samples_ms()     # HIP events around the assembly kernel of every call, on the context's stream
    kern_ms, n_calls = float(sum(samples)), len(samples)
    ctx.timing_enable(False)
    multi = None
    if world > 1:
        # diagnostics of the N > 1 run: the exchange alone (device time between events on this rank's stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fence()
        e0.record()
        for _ in range(a.steps):
            hx.exchange(u_t)
        e1.record()
        fence()
        halo_ms = e0.elapsed_time(e1) / a.steps
        on_cpu = a.backend != "nccl"
        t = torch.tensor([dt, kern_ms / max(n_calls, 1) if not overlap else kern_ms / a.steps, halo_ms, t_host / a.steps * 1e6,
                          float(hx.bytes_per_step), float(l_conn.shape[0]) / max(lp.n_elem_owned, 1), float(len(hx.peers))],
                         dtype=torch.float64, device="cpu" if on_cpu else dev)
        tmin = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        dt, kern_avg_ms = float(t[0]), float(t[1])
        multi = {"halo_ms_max": float(t[2]), "halo_ms_min": float(tmin[2]), "host_enqueue_us_per_step_max": float(t[3]),
                 "halo_send_bytes_per_rank_max": float(t[4]), "halo_send_bytes_per_rank_min": float(tmin[4]),
                 "local_over_owned_elements_max": float(t[5]), "local_over_owned_elements_min": float(tmin[5]),
                 "peers_max": int(t[6]), "kernel_ms_per_step_max": float(t[1]), "kernel_ms_per_step_min": float(tmin[1]),
                 "interior_node_fraction_rank0": lp.n_interior / max(lp.n_owned, 1),
                 "note": "halo_ms = grouped isend/irecv round alone, back to back; kernel_ms = HIP events around the assembly "
                         "kernel(s) of a step; local_over_owned_elements = (partition + ghost layer) / partition"}
    else:
        kern_avg_ms = kern_ms / max(n_calls, 1)

    if rank == 0:
        b_alg = algorithmic_bytes(4, l_conn.shape[0], l_xyz.shape[0], n_owned, 5, 5, nnz)
        achieved = b_alg / (kern_avg_ms * 1e-3) / 1e9
        traffic, flop, prof_kernel = profile_numbers(a.n, world) if a.params == "shipped" and not a.opt else (None, None, None)
        fp64_tflops = flop / (kern_avg_ms * 1e-3) / 1e12 if flop else None
        out = {
            "metric": "elements_assembled_per_sec", "value": n_elem_global * a.steps / dt, "unit": "elements/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"PIHNA TET4 K({a.n}): {n_elem_global} tets, {n_node_global} nodes, 5 unknowns, "
                                   f"params run/PIHNA/input.dat ({a.params}), order={a.order}, state of SURVEY App. C (background "
                                   f"(0,0,0,7170,0) of the shipped field file, tumour values inside a sphere r = 0.25; see state_sensitivity)",
                       "scatter": ["auto", "coloured", "rowgather"][ctx.get_scatter()], "kernel_variant": a.variant, "options": a.opt,
                       "parallelism": (f"element partition x{world}, 1 ghost layer, halo p2p over " + ("RCCL" if a.backend == "nccl" else "gloo (host-staged rehearsal)") + (", overlapped with interior rows" if overlap else "")) if world > 1 else "single GPU",
                       "rank0_local_elements": int(l_conn.shape[0]), "rank0_nnz": int(nnz)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": b_alg, "kernel_ms_avg": kern_avg_ms,
                         "kernel_ms_median": float(np.median(samples)) if samples and world == 1 else None,
                         "kernel_ms_min": float(min(samples)) if samples and world == 1 else None, "launches_timed": len(samples),
                         "fp64_tflops": fp64_tflops, "fp64_peak_tflops": FP64_PEAK_TFLOPS,
                         "fp64_frac": fp64_tflops / FP64_PEAK_TFLOPS if fp64_tflops else None,
                         "profile_kernel": prof_kernel,
                         "note": "traffic / fp64_* come from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json) and are "
                                 "null unless that profile was taken from the kernel sources built now"},
        }
        if multi:
            out["multi_gpu"] = multi
        if world == 1:
            gpu_val = gpu_rhs = None
            if a.cpu_baseline:
                gpu_val, gpu_rhs = ctx.csr_download()     # the result of the last timed step
            if a.two_part:
                out["two_part_host"] = two_part_host_cost(ctx, p, n_owned)
            if a.state_check and a.params == "shipped" and not a.opt:
                out["state_sensitivity"] = state_sensitivity(ctx, p, l_xyz, u_t, b_alg)
            if a.handback:
                out["handback"] = handback(ctx, lambda: ctx.assemble_pihna(p), n_owned, nnz, n_rows, kern_avg_ms)
                # end to end when the adapter downloads the CSR every step (never `value`: inputs and outputs of the metric stay in HBM)
                hb = out["handback"]["kernel_plus_handback_overlapped_ms"] or out["handback"]["kernel_plus_handback_ms"]
                out["value_with_handback"] = n_elem_global / (hb * 1e-3)
    ctx.close()
    del u_t
    if rank == 0 and world == 1:
        torch.cuda.empty_cache()
        if a.cpu_baseline:
            out["cpu_baseline"], out["parity"] = cpu_baseline(l_conn, l_xyz, l_u, p, f"K({a.n})", a.cpu_sample, a.params, gpu_val, gpu_rhs)
            del gpu_val, gpu_rhs
        if a.configs:
            out["configs"] = extra_configs()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
