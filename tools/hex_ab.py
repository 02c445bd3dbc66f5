#!/usr/bin/env python3
"""HEX8 generic row gather: node-staged kernel (staged=1) vs registers-resident (staged=0), per model."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from rdcfes_amd import (AssemblyContext, adpm_params_from_dict, hcc_params_from_dict, pihna_params_from_dict,
                        proteas_params_from_dict, ripf_params_from_dict, synth)
from rdcfes_amd.context import FIELD_AUX_NODAL, FIELD_ELEM_FIBRE, FIELD_OLD_SOLUTION
n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
conn, xyz = synth.hex_mesh(n, jitter=0.1)
cases = {
    "hcc": (3, hcc_params_from_dict(synth.hcc_param_dict("full")), synth.hcc_fields(xyz), None, None, "assemble_hcc"),
    "ripf": (3, ripf_params_from_dict(synth.ripf_param_dict("shipped")), *synth.ripf_fields(xyz), None, "assemble_ripf"),
    "adpm": (3, adpm_params_from_dict(synth.adpm_param_dict("full"), time=3.0), *[synth.adpm_fields(xyz, conn.shape[0])[i] for i in (0,)], None,
             synth.adpm_fields(xyz, conn.shape[0])[1], "assemble_adpm"),
    "pihna": (5, pihna_params_from_dict(synth.pihna_param_dict("shipped")), synth.pihna_fields(xyz), None, None, "assemble_pihna"),
    "proteas": (5, proteas_params_from_dict(synth.proteas_param_dict("full")), *synth.proteas_fields(xyz), None, "assemble_proteas"),
}
for name, (nv, p, u, aux, ed, fn) in cases.items():
    for staged in (-1, -2, 2, 0):   # -1 = cluster kernel (default for three unknowns), -2 = its persistent form, 2 = force the node-staged pair kernel, 0 = registers-resident
        if staged == -2 and nv != 3: continue   # the persistent form exists for three unknowns only
        with AssemblyContext(0) as c:
            c.set_option("hex_kernel", {-1: 0, -2: 2}.get(staged, 1))
            c.set_option("staged", max(staged, 0))
            c.mesh_upload(8, conn, xyz, nv)
            c.field_upload(FIELD_OLD_SOLUTION, u)
            if aux is not None: c.field_upload(FIELD_AUX_NODAL, aux)
            if ed is not None: c.field_upload(FIELD_ELEM_FIBRE, ed)
            getattr(c, fn)(p); c.synchronize()
            c.timing_enable(True)
            for _ in range(3): getattr(c, fn)(p)
            ms, cnt = c.timing_sum_ms()
            print(f"{name:8s} H({n}) staged={staged}: {ms / cnt:8.3f} ms", flush=True)
