#!/usr/bin/env python3
"""Parity of a tuning-kernel selection (rdc_set_option) against the oracle on small meshes."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import oracle as O
from rdcfes_amd import AssemblyContext, pihna_params_from_dict, ripf_params_from_dict, hcc_params_from_dict, synth
from rdcfes_amd.context import FIELD_OLD_SOLUTION, FIELD_AUX_NODAL
opts = dict(kv.split("=") for kv in sys.argv[1:])
worst = 0.0
for model, variant, n, order in ((0, "shipped", 9, "lex"), (0, "full", 7, "random"), (1, "full", 7, "random"), (2, "full", 8, "lex"), (0, "shipped", 24, "lex")):
    conn, xyz = synth.kuhn_tet_mesh(n, order=order)
    aux = None
    if model == 0: p, u, nv = pihna_params_from_dict(synth.pihna_param_dict(variant)), synth.pihna_fields(xyz), 5
    elif model == 1:
        p, nv = ripf_params_from_dict(synth.ripf_param_dict(variant)), 3
        u, aux = synth.ripf_fields(xyz)
    else: p, u, nv = hcc_params_from_dict(synth.hcc_param_dict(variant)), synth.hcc_fields(xyz), 3
    for frac in (1.0, 0.6):
        n_owned = int(frac * xyz.shape[0])
        c2 = conn[(conn < n_owned).any(axis=1)]
        _, _, val0, rhs0 = O.assemble(model, 4, c2, xyz, nv, p, u_old=u, aux=aux, n_owned=n_owned)
        with AssemblyContext(0) as ctx:
            ctx.mesh_upload(4, c2, xyz, nv, n_owned=n_owned)
            ctx.field_upload(FIELD_OLD_SOLUTION, u)
            if aux is not None: ctx.field_upload(FIELD_AUX_NODAL, aux)
            for k, v in opts.items(): ctx.set_option(k, int(v))
            for rep in range(2):
                [ctx.assemble_pihna, ctx.assemble_ripf, ctx.assemble_hcc][model](p)
                val, rhs = ctx.csr_download()
                e = max(np.linalg.norm(val - val0) / np.linalg.norm(val0), np.linalg.norm(rhs - rhs0) / np.linalg.norm(rhs0))
                worst = max(worst, e)
                print(f"model {model} {variant} K({n}) {order} owned {frac} rep {rep}: {e:.2e}", flush=True)
assert worst < 1e-10, worst
print("OK", worst)
