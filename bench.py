#!/usr/bin/env python3
"""bench.py — elements assembled per second of the PIHNA element-assembly hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one invocation of the assemble callback over the whole mesh: (N > 1: halo exchange of
ghost-node DoFs over RCCL, then) evaluation of Ke/Fe for every element and scatter into the global
CSR matrix + rhs.  Inputs are resident in HBM when the timed region starts.

Workload: BASELINE.json's metric is quoted on the 10M-tet PIHNA mesh, which fits one GPU:
K(119) = 10,110,954 TET4 / 1,728,000 nodes, 5 unknowns (8.64 M DoFs, 648 M CSR values), synthetic
fields, parameters of run/PIHNA/input.dat.  N > 1 partitions that SAME mesh (strong scaling).
Prints ONE JSON line on rank 0.
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

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes(nen, n_elem, n_node, n_owned, nvar, n_in, nnz):
    """SURVEY §8d / BASELINE.md §3: compulsory traffic of one assembly pass (FP64 values, int32 ids)."""
    return 4 * nen * n_elem + 8 * 3 * n_node + 8 * n_in * n_node + 8 * nnz + 8 * nvar * n_owned


def cpu_baseline(n_sample, param_variant):
    """Oracle ("port" of the reference loop + MatSetValues-like insertion) on 1 host core, bounded sample."""
    from oracle import oracle as O
    from rdcfes_amd import pihna_params_from_dict, synth
    conn, xyz = synth.kuhn_tet_mesh(n_sample, order="lex")
    u = synth.pihna_fields(xyz)
    p = pihna_params_from_dict(synth.pihna_param_dict(param_variant))
    pattern = O.build_pattern(4, conn, xyz.shape[0], xyz.shape[0], 5)[:2]
    t0 = time.perf_counter()
    O.assemble(O.MODEL_PIHNA, 4, conn, xyz, 5, p, u_old=u, pattern=pattern)
    dt = time.perf_counter() - t0
    return {"value": conn.shape[0] / dt, "unit": "elements/s", "cores": 1, "kind": "port",
            "sample": f"K({n_sample}) = {conn.shape[0]} TET4 of the same generator/fields/params, full assembly "
                      f"incl. sorted-row CSR insertion, {dt:.1f} s on 1 core (oracle/rdc_oracle.c, gcc -O2)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
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
    ap.add_argument("--cpu-sample", type=int, default=84, help="K(m) sample for the CPU baseline (0 = skip)")
    a = ap.parse_args()

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
    # the exchange on its stream (tools/two_part_ab.py: the split itself costs ~13 us per step at per-GPU size)
    overlap = hx is not None and a.overlap and lp.n_interior > 0
    if overlap:
        main_s, halo_s = torch.cuda.current_stream(), torch.cuda.Stream(device=dev)
        ctx.set_option("interior_nodes", int(lp.n_interior))

    def step():
        if overlap:
            halo_s.wait_stream(main_s)      # the previous step has read the ghost rows this exchange overwrites
            ctx.set_option("part", 1)
            ctx.assemble_pihna(p)           # interior rows, main stream, concurrent with the exchange
            with torch.cuda.stream(halo_s):
                hx.exchange(u_t)
            ctx.set_stream(halo_s.cuda_stream)
            ctx.set_option("part", 2)
            ctx.assemble_pihna(p)           # rows next to ghosts, behind the exchange on its stream (fills part 1's tail)
            ctx.set_stream(main_s.cuda_stream)
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
    fence()
    dt = time.perf_counter() - t0
    kern_ms, n_calls = ctx.timing_sum_ms()
    ctx.timing_enable(False)
    if world > 1:
        t = torch.tensor([dt, kern_ms / max(n_calls, 1)], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        if overlap:
            t[1] = kern_ms / a.steps        # two launches per step
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, kern_avg_ms = float(t[0]), float(t[1])
    else:
        kern_avg_ms = kern_ms / max(n_calls, 1)

    if rank == 0:
        b_alg = algorithmic_bytes(4, l_conn.shape[0], l_xyz.shape[0], n_owned, 5, 5, nnz)
        achieved = b_alg / (kern_avg_ms * 1e-3) / 1e9
        traffic = None
        pmc = ROOT / "profiles" / "pmc_traffic.json"  # written from rocprofv3 --pmc passes, see profiles/README.md
        if pmc.exists():
            try:
                t = json.loads(pmc.read_text())
                if t.get("workload") == f"K({a.n})" and t.get("n_gpus") == world:
                    traffic = t.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "elements_assembled_per_sec", "value": n_elem_global * a.steps / dt, "unit": "elements/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"PIHNA TET4 K({a.n}): {n_elem_global} tets, {n_node_global} nodes, 5 unknowns, "
                                   f"params run/PIHNA/input.dat ({a.params}), order={a.order}",
                       "scatter": ["auto", "coloured", "rowgather"][ctx.get_scatter()], "kernel_variant": a.variant, "options": a.opt,
                       "parallelism": (f"element partition x{world}, 1 ghost layer, halo p2p over " + ("RCCL" if a.backend == "nccl" else "gloo (host-staged rehearsal)") + (", overlapped with interior rows" if overlap else "")) if world > 1 else "single GPU",
                       "rank0_local_elements": int(l_conn.shape[0]), "rank0_nnz": int(nnz)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": b_alg, "kernel_ms_avg": kern_avg_ms},
        }
        if world == 1 and a.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(a.cpu_sample, a.params)
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
