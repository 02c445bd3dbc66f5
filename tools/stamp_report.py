#!/usr/bin/env python3
"""Phase breakdown of k_tet4_rg3 from in-kernel s_memtime stamps (diagnostic build; shares, not absolute times)."""
import ctypes as C, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from rdcfes_amd import AssemblyContext, pihna_params_from_dict, synth
from rdcfes_amd.context import FIELD_OLD_SOLUTION
n = int(sys.argv[1]) if len(sys.argv) > 1 else 119
conn, xyz = synth.kuhn_tet_mesh(n)
p, u = pihna_params_from_dict(synth.pihna_param_dict("shipped")), synth.pihna_fields(xyz)
ctx = AssemblyContext(0)
ctx.mesh_upload(4, conn, xyz, 5); ctx.field_upload(FIELD_OLD_SOLUTION, u)
for _ in range(3): ctx.assemble_pihna(p)
ctx.synchronize()
nw = C.c_int64()
ctx._ck(ctx._lib.rdc_debug_stamps(ctx._h, None, 0, C.byref(nw)))
ctx.assemble_pihna(p); ctx.synchronize()
buf = np.zeros(nw.value, dtype=np.int64)
ctx._ck(ctx._lib.rdc_debug_stamps(ctx._h, buf.ctypes.data_as(C.POINTER(C.c_longlong)), buf.size, C.byref(nw)))
nwg = buf.size // 36
t = buf[:nwg * 24].reshape(-1, 4, 6).astype(np.float64)
sub = buf[nwg * 24:].reshape(-1, 4, 3).astype(np.float64)
d = np.diff(t, axis=2)  # [wg][wave][5 phases]
names = ["loads+zero+barrier", "compute (prepare+rows+atomics issue)", "wait at barrier (LDS drain, slowest wave)", "fold + barrier", "flush stores issue"]
tot = (t[:, :, 5] - t[:, :, 0])
print(f"workgroups {t.shape[0]}, wave lifetime median {np.median(tot):.0f} cycles (100 MHz-domain ticks? see note), mean {tot.mean():.0f}")
for i, nm in enumerate(names):
    print(f"{nm:45s} median {np.median(d[:, :, i]):8.0f}  mean {d[:, :, i].mean():8.0f}  share {d[:, :, i].sum() / tot.sum():6.1%}")
span = t[:, :, 5].max() - t[:, :, 0].min()
print("kernel span (ticks):", span)
print("phase 0 detail (cycles since wave start): level-1 loads back", np.median(sub[:, :, 0]), " DMA issued + ntab", np.median(sub[:, :, 1]),
      " zero done + DMA landed", np.median(sub[:, :, 2]), " (waves with a DMA round:", np.median(sub[:, :2, 2]), " others:", np.median(sub[:, 2:, 2]), ")")
