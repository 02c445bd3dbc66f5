#!/usr/bin/env python3
"""Fused cluster kernel vs the two-pass solid assembly (and its diagnostic options) on H(n), interleaved rounds."""
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
    sel = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else None
    for rnd in range(2):
      for idx, (name, opts) in enumerate((("fused 3+1 waves (default)", {}), ("fused 6+2 waves", {"solid_cl_waves": 62}), ("two-pass", {"solid_kernel": 2}),
                         ("two-pass, pass 1 without stores", {"solid_kernel": 2, "solid_store": 2}), ("fused 3+1, node-distinct pair order", {"solid_cl_order": 0}),
                         # timing diagnostics of k_solid_cl (results wrong by construction): "solid_store" bit mask
                         # 1 = consumers idle, 2 = producers idle, 4 = image not zeroed, 8 = no atomics, 16 = no copy-out, 32 = no element loads
                         ("fused, consumers idle", {"solid_store": 1}), ("fused, producers idle", {"solid_store": 2}), ("fused, no compute", {"solid_store": 3}),
                         ("fused, no compute, no atomics", {"solid_store": 11}), ("fused, no compute, no copy-out", {"solid_store": 19}),
                         ("fused, no compute, no element loads", {"solid_store": 35}), ("fused, barriers + lists only", {"solid_store": 63}),
                         ("fused, compute only", {"solid_store": 28}), ("fused, copy-out with 8-byte stores", {"solid_store": 64}))):
        if sel is not None and idx not in sel: continue
        c.set_option("solid_store", 0); c.set_option("solid_split", 1); c.set_option("solid_gather", 0); c.set_option("solid_kernel", 0); c.set_option("solid_cl_waves", 31); c.set_option("solid_cl_order", 1)
        for k, v in opts.items(): c.set_option(k, v)
        c.solid_assemble(sp, True); c.synchronize()
        c.timing_enable(True)
        for _ in range(3): c.solid_assemble(sp, True)
        ms, cnt = c.timing_sum_ms()
        c.timing_enable(False)
        print(f"{name:34s} {ms / cnt:8.3f} ms", flush=True)
