#!/usr/bin/env python3
"""Seeded stress run of the round-2 kernels against the threaded oracle (GPU box): HEX8 HCC cluster kernel, fused solid kernel
(with and without use_symmetry), TET4 RIPF element visits, PIHNA moments with mirror blocks; jittered meshes, lexicographic and
random numbering, ghosted partitions.  python tools/stress_parity.py"""
import sys, numpy as np
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import oracle as O
O.build()
from rdcfes_amd import (AssemblyContext, SolidMaterial, SolidParams, hcc_params_from_dict, ripf_params_from_dict, pihna_params_from_dict, synth)
from rdcfes_amd.context import FIELD_AUX_NODAL, FIELD_ELEM_FIBRE, FIELD_OLD_SOLUTION, FIELD_UNDEFORMED_XYZ
def rel(a, b): return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)
worst = 0.0
for seed in (1, 2, 3):
    for order in ("lex", "random"):
        rng = np.random.default_rng(seed)
        # HEX8: HCC (cluster kernel) and solid (fused kernel), with a ghosted partition
        n = 18 + seed
        conn, Xu = synth.hex_mesh(n, jitter=0.2, seed=seed, order=order)
        n_owned = int((0.5 + 0.1 * seed) * Xu.shape[0]) if order == "random" else Xu.shape[0]
        keep = (conn < n_owned).any(axis=1); conn = conn[keep]
        u = rng.uniform(0.0, 0.3, (Xu.shape[0], 3))
        p = hcc_params_from_dict(synth.hcc_param_dict("full"))
        _, _, v0, r0 = O.assemble(O.MODEL_HCC, 8, conn, Xu, 3, p, u_old=u, n_owned=n_owned, threads=8)
        with AssemblyContext(0) as c:
            c.mesh_upload(8, conn, Xu, 3, n_owned=n_owned); c.field_upload(FIELD_OLD_SOLUTION, u); c.assemble_hcc(p); v, r = c.csr_download()
        e = max(rel(v, v0), rel(r, r0)); worst = max(worst, e); print("hcc hex", seed, order, n, e, flush=True)
        x = Xu + synth.solid_displacement(Xu, amp=0.01)
        em = rng.integers(0, 2, conn.shape[0]).astype(np.int32)
        mats = [SolidMaterial(2.0e3, 0.4, 0.0, (0.0, 0.0, 0.0)), SolidMaterial(1.5e3, 0.35, 40.0, (0.3, 0.2, 0.1))]
        fibre = rng.standard_normal((conn.shape[0], 3))
        for sym in (0, 1):
            sp = SolidParams(0.4, 1.0e5, sym, 0)
            _, _, v0, r0 = O.assemble(O.MODEL_SOLID, 8, conn, x, 3, sp, xyz_undeformed=Xu, elem_fibre=fibre, elem_material=em, materials=mats, request_jacobian=True, n_owned=n_owned, threads=8)
            with AssemblyContext(0) as c:
                c.mesh_upload(8, conn, x, 3, n_owned=n_owned); c.field_upload(FIELD_UNDEFORMED_XYZ, Xu); c.field_upload(FIELD_ELEM_FIBRE, fibre); c.solid_set_materials(em, mats)
                c.solid_assemble(sp, True); v, r = c.csr_download()
            e = max(rel(v, v0), rel(r, r0)); worst = max(worst, e); print("solid hex sym", sym, seed, order, n, e, flush=True)
        # TET4: RIPF all terms (element visits, coefficient form) and PIHNA shipped (moments + mirror blocks), ghosted
        n = 14 + seed
        conn, xyz = synth.kuhn_tet_mesh(n, jitter=0.2, seed=seed, order=order)
        n_owned = int((0.5 + 0.1 * seed) * xyz.shape[0]) if order == "random" else xyz.shape[0]
        keep = (conn < n_owned).any(axis=1); conn = conn[keep]
        u, aux = synth.ripf_fields(xyz, seed=seed)
        p = ripf_params_from_dict(synth.ripf_param_dict("full"))
        _, _, v0, r0 = O.assemble(O.MODEL_RIPF, 4, conn, xyz, 3, p, u_old=u, aux=aux, n_owned=n_owned, threads=8)
        with AssemblyContext(0) as c:
            c.mesh_upload(4, conn, xyz, 3, n_owned=n_owned); c.field_upload(FIELD_OLD_SOLUTION, u); c.field_upload(FIELD_AUX_NODAL, aux); c.assemble_ripf(p); v, r = c.csr_download()
        e = max(rel(v, v0), rel(r, r0)); worst = max(worst, e); print("ripf tet", seed, order, n, e, flush=True)
        u = synth.pihna_fields(xyz, seed=seed)
        for pv in ("shipped", "realexp"):
            p = pihna_params_from_dict(synth.pihna_param_dict(pv))
            _, _, v0, r0 = O.assemble(O.MODEL_PIHNA, 4, conn, xyz, 5, p, u_old=u, n_owned=n_owned, threads=8)
            with AssemblyContext(0) as c:
                c.mesh_upload(4, conn, xyz, 5, n_owned=n_owned); c.field_upload(FIELD_OLD_SOLUTION, u); c.assemble_pihna(p); v, r = c.csr_download()
            e = max(rel(v, v0), rel(r, r0)); worst = max(worst, e); print("pihna tet", pv, seed, order, n, e, flush=True)
print("WORST", worst)
assert worst < 1e-10
