#!/usr/bin/env python3
"""A/B harness: one process, one mesh, interleaved rounds over tuning-option sets (guide rule 24).
    python tools/ab.py --n 119 --rounds 5 "occupancy=1" "occupancy=2" "occupancy=2,ablate=1" ...
"""
import argparse, json, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from rdcfes_amd import AssemblyContext, pihna_params_from_dict, ripf_params_from_dict, hcc_params_from_dict, synth
from rdcfes_amd.context import FIELD_OLD_SOLUTION, FIELD_AUX_NODAL

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=119)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--params", default="shipped")
ap.add_argument("--model", default="pihna")
ap.add_argument("--order", default="lex")
ap.add_argument("--scatter", type=int, default=2)
ap.add_argument("--dense", type=int, default=0, help="PIHNA: tumour state at every node (no background elements)")
ap.add_argument("sets", nargs="+")
a = ap.parse_args()
conn, xyz = synth.kuhn_tet_mesh(a.n, order=a.order)
ctx = AssemblyContext(0)
if a.model == "pihna":
    p, u, aux, nv = pihna_params_from_dict(synth.pihna_param_dict(a.params)), synth.pihna_fields(xyz, radius=10.0 if a.dense else 0.25), None, 5
    run = ctx.assemble_pihna
elif a.model == "ripf":
    p, nv = ripf_params_from_dict(synth.ripf_param_dict(a.params)), 3
    u, aux = synth.ripf_fields(xyz)
    run = ctx.assemble_ripf
elif a.model == "adpm":
    from rdcfes_amd import adpm_params_from_dict
    from rdcfes_amd.context import FIELD_ELEM_FIBRE
    p, nv, aux = adpm_params_from_dict(synth.adpm_param_dict("full"), time=3.0), 3, None
    u, tracts = synth.adpm_fields(xyz, conn.shape[0])
    run = ctx.assemble_adpm
elif a.model == "proteas":
    from rdcfes_amd import proteas_params_from_dict
    p, nv = proteas_params_from_dict(synth.proteas_param_dict("full")), 5
    u, aux = synth.proteas_fields(xyz)
    run = ctx.assemble_proteas
else:
    p, u, aux, nv = hcc_params_from_dict(synth.hcc_param_dict(a.params)), synth.hcc_fields(xyz), None, 3
    run = ctx.assemble_hcc
ctx.mesh_upload(4, conn, xyz, nv)
ctx.field_upload(FIELD_OLD_SOLUTION, u)
if aux is not None:
    ctx.field_upload(FIELD_AUX_NODAL, aux)
if a.model == "adpm":
    ctx.field_upload(FIELD_ELEM_FIBRE, tracts)
ctx.set_scatter(a.scatter)
ctx.timing_enable(True)
res = {s: [] for s in a.sets}
cur_block = 2561
for r in range(a.rounds):
    for s in a.sets:
        ctx.set_option("occupancy", 2); ctx.set_option("ablate", 0); ctx.set_option("kernel", 0); ctx.set_option("specialise", 1); ctx.set_option("slim", 0); ctx.set_option("moments", 1); ctx.set_option("prefetch", 0); ctx.set_option("stagger", 0); ctx.set_option("lds_pad", 0); ctx.set_option("grid", 0); ctx.set_option("ev_occupancy", 3); ctx.set_option("evc_occupancy", 2); ctx.set_option("ev_resident", 1); ctx.set_option("xcd", 0); ctx.set_option("ev_general", 1); ctx.set_option("ev_background", 1)
        opts = dict(kv.split("=") for kv in s.split(",") if "=" in kv)
        blk = int(opts.pop("block", 256)) * 10 + int(opts.pop("schedule", 1))
        if blk != cur_block:  # work lists depend on the workgroup size / schedule: rebuild
            ctx.set_option("block", blk // 10)
            ctx.set_option("schedule", blk % 10)
            ctx.mesh_upload(4, conn, xyz, nv)
            ctx.field_upload(FIELD_OLD_SOLUTION, u)
            if aux is not None:
                ctx.field_upload(FIELD_AUX_NODAL, aux)
            ctx.set_scatter(a.scatter)
            cur_block = blk
        for k, v in opts.items():
            ctx.set_option(k, int(v))
        run(p); ctx.synchronize(); ctx.timing_sum_ms()
        for _ in range(a.reps):
            run(p)
        ms, n = ctx.timing_sum_ms()
        res[s].append(ms / n)
ne = conn.shape[0]
for s in a.sets:
    v = np.array(res[s])
    print(f"{s:40s} median {np.median(v):8.3f} ms  min {v.min():8.3f} ms  -> {ne/np.median(v)/1e3:8.1f} Melem/s", flush=True)
