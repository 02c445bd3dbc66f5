#!/usr/bin/env python3
"""Pass-1 / pass-2 timing of the two-pass solid assembly under diagnostic options (tools/perf_table.py set-up)."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from rdcfes_amd import AssemblyContext, SolidMaterial, SolidParams, synth
from rdcfes_amd.context import FIELD_ELEM_FIBRE, FIELD_UNDEFORMED_XYZ
n = int(sys.argv[1]) if len(sys.argv) > 1 else 126
conn, Xu = synth.hex_mesh(n, jitter=0.1)
x = Xu + synth.solid_displacement(Xu)
em = (np.linalg.norm(Xu[conn].mean(axis=1) - 0.5, axis=1) < 0.3).astype(np.int32)
mats = [SolidMaterial(2.0e3, 0.4, 0.0, (0.0, 0.0, 0.0)), SolidMaterial(2.0e3, 0.4, 0.0, (0.3, 0.3, 0.3))]
sp = SolidParams(0.4, 1.0e8, 0, 0)
with AssemblyContext(0) as c:
    c.mesh_upload(8, conn, x, 3)
    c.field_upload(FIELD_UNDEFORMED_XYZ, Xu); c.field_upload(FIELD_ELEM_FIBRE, np.tile([0.0, 0.0, 1.0], (conn.shape[0], 1)))
    c.solid_set_materials(em, mats)
    for name, opts in (("default", {}), ("gather=1 (direct stores)", {"solid_gather": 1}), ("pass 1 direct stores", {"solid_store": 1}),
                       ("pass 1 without stores", {"solid_store": 2}), ("column split", {"solid_split": 0})):
        c.set_option("solid_store", 0); c.set_option("solid_split", 1); c.set_option("solid_gather", 0)
        for k, v in opts.items(): c.set_option(k, v)
        c.solid_assemble(sp, True); c.synchronize()
        c.timing_enable(True)
        for _ in range(3): c.solid_assemble(sp, True)
        ms, cnt = c.timing_sum_ms()
        c.timing_enable(False)
        print(f"{name:24s} {ms / cnt:8.3f} ms (pass 1 + gather + rhs)", flush=True)
