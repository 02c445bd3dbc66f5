#!/usr/bin/env python3
"""HEX8 cluster kernels on H(n): default vs persistent form vs pair kernel, and the timing diagnostics of the persistent form
(k_hex8_clp): "ablate" bit mask
1 = consumers idle, 2 = producers idle, 8 = no atomics, 16 = no copy-out."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from rdcfes_amd import AssemblyContext, hcc_params_from_dict, synth
from rdcfes_amd.context import FIELD_OLD_SOLUTION
n = int(sys.argv[1]) if len(sys.argv) > 1 else 126
params = sys.argv[2] if len(sys.argv) > 2 else "full"
conn, xyz = synth.hex_mesh(n, jitter=0.1)
p = hcc_params_from_dict(synth.hcc_param_dict(params))
u = synth.hcc_fields(xyz)
with AssemblyContext(0) as c:
    c.mesh_upload(8, conn, xyz, 3)
    c.field_upload(FIELD_OLD_SOLUTION, u)
    for rnd in range(2):
        for name, hk, ab in (("persistent", 2, 0), ("one workgroup per cluster (default)", 0, 0), ("pair kernel (staged)", 1, 0), ("consumers idle", 2, 1), ("producers idle", 2, 2),
                             ("no compute", 2, 3), ("no atomics", 2, 8), ("no copy-out", 2, 16), ("no compute, no atomics", 2, 11), ("no compute, no copy-out", 2, 19),
                             ("barriers and loads only", 2, 27), ("compute only", 2, 24), ("element-major pair order, persistent", 2, 0), ("element-major, one workgroup per cluster", 0, 0), ("one point per round", 0, 0)):
            c.set_option("hex_kernel", hk); c.set_option("ablate", ab); c.set_option("prefetch", 1 if name.startswith("one point") else 0); c.set_option("solid_cl_order", 1 if name.startswith("element-major") else 0)
            c.assemble_hcc(p); c.synchronize()
            c.timing_enable(True)
            for _ in range(3): c.assemble_hcc(p)
            ms, cnt = c.timing_sum_ms()
            c.timing_enable(False)
            print(f"{name:32s} {ms / cnt:8.3f} ms", flush=True)
