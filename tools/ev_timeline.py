#!/usr/bin/env python3
"""Per-wave timeline of k_tet4_ev from in-kernel s_memtime stamps (the stamped build computes the right matrix:
"kernel" = 7, "ablate" = 4).  Prints where a workgroup's lifetime goes and how many workgroups a CU holds in each phase.
    python tools/ev_timeline.py [n=119] [nodrain]      nodrain: do not wait for the stores at the end (stamp 10 == stamp 9)
"""
import ctypes as C, json, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from rdcfes_amd import AssemblyContext, pihna_params_from_dict, synth
from rdcfes_amd.context import FIELD_OLD_SOLUTION

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 119
nodrain = "nodrain" in sys.argv
evq = "evq" in sys.argv      # the pipelined resident kernel ("ev_resident" = 1): 6 stamps per cluster and wave
conn, xyz = synth.kuhn_tet_mesh(n)
p, u = pihna_params_from_dict(synth.pihna_param_dict("shipped")), synth.pihna_fields(xyz)
ctx = AssemblyContext(0)
ctx.mesh_upload(4, conn, xyz, 5); ctx.field_upload(FIELD_OLD_SOLUTION, u)
ctx.timing_enable(True)
for _ in range(3): ctx.assemble_pihna(p)
ctx.synchronize(); ms, k = ctx.timing_sum_ms(); base_ms = ms / k
val0, rhs0 = ctx.csr_download()
ctx.set_option("kernel", 7); ctx.set_option("ablate", 4)
ctx.set_option("ev_resident", 2 if evq else 0)
if nodrain: ctx.set_option("stagger", -1)
nw = C.c_int64()
ctx._ck(ctx._lib.rdc_debug_stamps(ctx._h, None, 0, C.byref(nw)))
ctx.assemble_pihna(p); ctx.synchronize(); ms, k = ctx.timing_sum_ms()
val, rhs = ctx.csr_download()
buf = np.zeros(nw.value, dtype=np.int64)
ctx._ck(ctx._lib.rdc_debug_stamps(ctx._h, buf.ctypes.data_as(C.POINTER(C.c_longlong)), buf.size, C.byref(nw)))
print(f"K({n}): default kernel {base_ms:.3f} ms; stamped build {ms / k:.3f} ms; results equal to the default's: "
      f"{np.array_equal(val, val0) or float(np.abs(val - val0).max() / np.abs(val0).max())}")
if evq:
    t = buf.reshape(-1, 4, 12)[:, :, :6].astype(np.float64)
    t = t[t[:, 0, 0] > 0]
    d = np.diff(t, axis=2)
    life = t[:, :, 5] - t[:, :, 0]
    print(f"clusters {t.shape[0]}; per-cluster time of a wave: mean {life.mean():.0f} median {np.median(life):.0f} ticks")
    for i, nm in enumerate(["0 zero + wait for the fetched lists / records and the previous stores", "1 barrier", "2 list / record reads + visits",
                            "3 barrier (slowest wave)", "4 fetch issue + moment reads + expansion + image halves + copy-out"]):
        x = d[:, :, i]
        print(f"  {nm:72s} mean {x.mean():8.0f}  median {np.median(x):8.0f}  p90 {np.percentile(x, 90):8.0f}  share {x.sum() / life.sum():6.1%}")
    sys.exit(0)
t = buf.reshape(-1, 4, 12)
hw = t[:, :, 11]
t = t[:, :, :11].astype(np.float64)
ok = t[:, 0, 0] > 0
t, hw = t[ok], hw[ok]
names = ["0 zero + list loads + record DMA issued (first round trip)", "1 record DMA lands (second round trip)", "2 barrier",
         "3 visits: arithmetic + own LDS atomics", "4 barrier (slowest wave)", "5 moment reads + rhs stores", "6 barrier + expansion + image",
         "7 barrier", "8 copy-out: LDS reads + stores issued", "9 stores acknowledged"]
d = np.diff(t, axis=2)
life = t[:, :, 10] - t[:, :, 0]
print(f"workgroups {t.shape[0]}; wave lifetime mean {life.mean():.0f} median {np.median(life):.0f} ticks")
for i, nm in enumerate(names):
    x = d[:, :, i]
    print(f"  {nm:62s} mean {x.mean():8.0f}  median {np.median(x):8.0f}  p90 {np.percentile(x, 90):8.0f}  share {x.sum() / life.sum():6.1%}")
# (the s_memtime counters of different XCDs are not synchronised: only differences inside a wave are used)
out = {"n": n, "default_ms": base_ms, "stamped_ms": ms / k, "mean_ticks": {names[i]: float(d[:, :, i].mean()) for i in range(10)},
       "lifetime_mean": float(life.mean())}
Path("gpurun_out").mkdir(exist_ok=True)
Path("gpurun_out/ev_timeline.json").write_text(json.dumps(out, indent=1))
