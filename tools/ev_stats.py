#!/usr/bin/env python3
"""Element-visit list statistics on the CPU (no GPU): r histogram, lane fill, LDS atomic wave-instructions per cluster.
    python tools/ev_stats.py --n 40
"""
import argparse, ctypes as C, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import build_shim
from rdcfes_amd import synth
ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=40); ap.add_argument("--order", default="lex")
ap.add_argument("--budget", type=int, default=53 * 1024)
a = ap.parse_args()
shim = build_shim()
conn, xyz = synth.kuhn_tet_mesh(a.n, order=a.order)
conn = np.ascontiguousarray(conn, dtype=np.uint32)
rc = shim.shim_prep_build(4, C.c_int64(conn.shape[0]), C.c_int64(xyz.shape[0]), C.c_int64(xyz.shape[0]), conn.ctypes.data_as(C.POINTER(C.c_uint32)), 5, C.c_int64(60 * 1024), 256)
assert rc == 0
st = (C.c_int64 * 6)()
assert shim.shim_ev_build(C.c_int64(a.budget), st) == 0, shim.shim_prep_error()
nwg, nvis, nrows, nls, maxout, cov = list(st)
vloc = np.empty(shim.shim_prep_size(30), dtype=np.uint32); shim.shim_prep_copy(30, vloc.ctypes.data_as(C.c_void_p))
desc = np.empty(shim.shim_prep_size(31) // 32, dtype=np.dtype([("nown","<u4"),("nvis","<u4"),("ntouch","<u4"),("nb","<u4"),("out","<u4"),("mn","<u4"),("mx","<u4"),("pad","<u4")]))
shim.shim_prep_copy(31, desc.ctypes.data_as(C.c_void_p))
vloc = vloc.reshape(nwg, 256)
valid = vloc != 0xFFFFFFFF
li = np.stack([(vloc >> (8 * j)) & 0xFF for j in range(4)], axis=-1)
r = (li < desc["nown"][:, None, None]).sum(-1) * valid
print(f"K({a.n}): {conn.shape[0]} elems, {nwg} clusters, nodes/cluster {desc['nown'].mean():.2f}, visits/cluster {nvis/nwg:.1f}, visits/elem {nvis/conn.shape[0]:.3f}, rows/visit {nrows/nvis:.3f}, nls {nls}, blocks/cluster {desc['nb'].mean():.1f}, ntouch {desc['ntouch'].mean():.1f}")
h = np.bincount(r[valid].ravel(), minlength=5)
print("r histogram (fraction of visits):", (h / h.sum()).round(3)[1:])
# wave-level instruction count: position i issued by a wave if any lane has r > i; cost 15(4-i)+9
rw = r.reshape(nwg, 4, 64)
cost = np.array([69, 54, 39, 24])
issued = np.stack([(rw > i).any(-1) for i in range(4)], -1)       # [nwg][wave][pos]
lanes = np.stack([(rw > i).sum(-1) for i in range(4)], -1)
ins = (issued * cost).sum((1, 2))
ideal = (lanes * cost).sum((1, 2)) / 64
print(f"atomic wave-instructions / cluster: {ins.mean():.1f}  (lane-adds/64 = {ideal.mean():.1f}, utilisation {ideal.mean()/ins.mean():.3f})")
for i in range(4):
    print(f"  position {i}: issued by {issued[:,:,i].sum(1).mean():.2f} waves/cluster, mean active lanes when issued {lanes[:,:,i][issued[:,:,i]].mean():.1f}")
print("waves with any visit / cluster:", valid.reshape(nwg,4,64).any(-1).sum(1).mean())
# group-level (16 lanes) passes
rg = r.reshape(nwg, 16, 16)
print("group passes (sum max r per group)/cluster:", rg.max(-1).sum(1).mean(), " rows/cluster:", r.sum((1)).mean())
print("host statistics: collisions", shim.shim_prep_size(33), " pass instructions / cluster", shim.shim_prep_size(34) / nwg)
