#!/usr/bin/env python3
"""Copy-out of the HEX8 cluster kernels with 16-byte (default) against 8-byte non-temporal stores on H(n): HCC (all terms,
shipped parameters) and the fused solid tangent; interleaved rounds, median."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from rdcfes_amd import AssemblyContext, hcc_params_from_dict, synth
from rdcfes_amd.context import FIELD_OLD_SOLUTION
n = int(sys.argv[1]) if len(sys.argv) > 1 else 126
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
conn, xyz = synth.hex_mesh(n, jitter=0.1)
u = synth.hcc_fields(xyz)
res = {}
with AssemblyContext(0) as c:
    c.mesh_upload(8, conn, xyz, 3)
    c.field_upload(FIELD_OLD_SOLUTION, u)
    for rnd in range(rounds):
        for params in ("full", "shipped"):
            p = hcc_params_from_dict(synth.hcc_param_dict(params))
            for ab in (0, 64):
                c.set_option("ablate", ab)
                c.assemble_hcc(p); c.synchronize()
                c.timing_enable(True)
                for _ in range(3): c.assemble_hcc(p)
                ms, cnt = c.timing_sum_ms()
                c.timing_enable(False)
                res.setdefault(f"HCC {params:8s} {'8-byte' if ab else '16-byte'}", []).append(ms / cnt)
    c.set_option("ablate", 0)
for k, v in res.items():
    print(f"{k:36s} median {np.median(v):7.3f} ms  min {min(v):7.3f} ms", flush=True)

# ---- fused solid tangent ------------------------------------------------------------------------------------------------
from rdcfes_amd import SolidMaterial, SolidParams
from rdcfes_amd.context import FIELD_ELEM_FIBRE, FIELD_UNDEFORMED_XYZ
Xu = xyz
x = Xu + synth.solid_displacement(Xu)
em = (np.linalg.norm(Xu[conn].mean(axis=1) - 0.5, axis=1) < 0.3).astype(np.int32)
mats = [SolidMaterial(2.0e3, 0.4, 0.0, (0.0, 0.0, 0.0)), SolidMaterial(2.0e3, 0.4, 0.0, (0.3, 0.3, 0.3))]
sp = SolidParams(0.4, 1.0e8, 0, 0)
res = {}
with AssemblyContext(0) as c:
    c.mesh_upload(8, conn, x, 3)
    c.field_upload(FIELD_UNDEFORMED_XYZ, Xu); c.field_upload(FIELD_ELEM_FIBRE, np.tile([0.0, 0.0, 1.0], (conn.shape[0], 1)))
    c.solid_set_materials(em, mats)
    for rnd in range(rounds):
        for st in ((64, 0) if rnd % 2 else (0, 64)):   # alternate which one goes first
            c.set_option("solid_store", st)
            c.solid_assemble(sp, True); c.synchronize()
            c.timing_enable(True)
            for _ in range(3): c.solid_assemble(sp, True)
            ms, cnt = c.timing_sum_ms()
            c.timing_enable(False)
            res.setdefault(f"solid tangent {'8-byte' if st else '16-byte'}", []).append(ms / cnt)
    c.set_option("solid_store", 0)
for k, v in res.items():
    print(f"{k:36s} median {np.median(v):7.3f} ms  min {min(v):7.3f} ms  all {' '.join(f'{t:.2f}' for t in v)}", flush=True)
