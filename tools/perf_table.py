#!/usr/bin/env python3
"""Per-configuration kernel timings (HIP events) for DESIGN.md: all models / element types of SURVEY §8."""
import json, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from rdcfes_amd import (AssemblyContext, SolidMaterial, SolidParams, adpm_params_from_dict, proteas_params_from_dict, hcc_params_from_dict, pihna_params_from_dict,
                        ripf_params_from_dict, synth)
from rdcfes_amd.context import FIELD_AUX_NODAL, FIELD_ELEM_FIBRE, FIELD_OLD_SOLUTION, FIELD_UNDEFORMED_XYZ

def b_alg(nen, ne, nn, nvar, n_in, nnz, solid=False):
    return 4 * nen * ne + 8 * 3 * nn * (2 if solid else 1) + 8 * n_in * nn + (8 * 4 * ne if solid else 0) + 8 * nnz + 8 * nvar * nn

def run(name, nen, conn, xyz, nv, setup, call, scatter, reps=10, n_in=None, solid=False, opts=()):
    with AssemblyContext(0) as ctx:
        for k, v in opts:
            ctx.set_option(k, v)
        t0 = time.time()
        ctx.mesh_upload(nen, conn, xyz, nv)
        prep = time.time() - t0
        setup(ctx)
        ctx.set_scatter(scatter)
        for _ in range(6):   # warm-up: the chip idles while the host prepares the mesh, the first launches run at low clocks
            call(ctx)
        ctx.synchronize()
        ctx.timing_enable(True)
        for _ in range(reps):
            call(ctx)
        ms, n = ctx.timing_sum_ms()
        ms /= n
        _, nnz = ctx.csr_dims()
        B = b_alg(nen, conn.shape[0], xyz.shape[0], nv, n_in if n_in is not None else nv, nnz, solid)
        print(json.dumps({"config": name, "scatter": ["auto", "coloured", "rowgather"][ctx.get_scatter()], "elements": int(conn.shape[0]),
                          "nodes": int(xyz.shape[0]), "nnz": int(nnz), "kernel_ms": round(ms, 3), "Melem_per_s": round(conn.shape[0] / ms / 1e3, 1),
                          "B_alg_per_elem": round(B / conn.shape[0], 1), "GBps_alg": round(B / ms / 1e6, 1), "frac_of_8TBps": round(B / ms / 1e6 / 8000, 4),
                          "colours": ctx.n_colours(), "prep_s": round(prep, 2)}), flush=True)

which = sys.argv[1:] or ["pihna55", "pihna119", "ripf94", "hcc_tet", "adpm94", "proteas94", "hcc126", "hcc126_shipped", "adpm100hex", "adpm94_shipped", "solid63", "solid126", "solid60tet", "pihna119_random"]
for w in which:
    if w.startswith("pihna"):
        order = "random" if w.endswith("random") else "lex"
        n = int(w.replace("pihna", "").replace("_random", ""))
        conn, xyz = synth.kuhn_tet_mesh(n, order=order)
        p, u = pihna_params_from_dict(synth.pihna_param_dict("shipped")), synth.pihna_fields(xyz)
        for sc in (2, 1) if n <= 60 else (2,):
            run(f"PIHNA TET4 K({n}) {order}", 4, conn, xyz, 5, lambda c: c.field_upload(FIELD_OLD_SOLUTION, u), lambda c: c.assemble_pihna(p), sc)
    elif w == "ripf94":
        conn, xyz = synth.kuhn_tet_mesh(94)
        p = ripf_params_from_dict(synth.ripf_param_dict("shipped"))
        u, aux = synth.ripf_fields(xyz)
        def setup(c):
            c.field_upload(FIELD_OLD_SOLUTION, u); c.field_upload(FIELD_AUX_NODAL, aux)
        run("RIPF TET4 K(94)", 4, conn, xyz, 3, setup, lambda c: c.assemble_ripf(p), 2, n_in=6)
    elif w == "hcc_tet":
        conn, xyz = synth.kuhn_tet_mesh(94)
        p, u = hcc_params_from_dict(synth.hcc_param_dict("full")), synth.hcc_fields(xyz)
        run("HCC TET4 K(94)", 4, conn, xyz, 3, lambda c: c.field_upload(FIELD_OLD_SOLUTION, u), lambda c: c.assemble_hcc(p), 2)
    elif w.startswith("proteas"):
        hexm = w.endswith("hex")
        n = int(w.replace("proteas", "").replace("hex", "") or 60)
        conn, xyz = synth.hex_mesh(n, jitter=0.1) if hexm else synth.kuhn_tet_mesh(n)
        u, aux = synth.proteas_fields(xyz)
        p = proteas_params_from_dict(synth.proteas_param_dict("full"))
        def setup(c):
            c.field_upload(FIELD_OLD_SOLUTION, u); c.field_upload(FIELD_AUX_NODAL, aux)
        run(f"PROTEAS {'HEX8 H' if hexm else 'TET4 K'}({n})", 8 if hexm else 4, conn, xyz, 5, setup, lambda c: c.assemble_proteas(p), 2, reps=6, n_in=6)
    elif w.startswith("adpm"):
        hexm = w.replace("_shipped", "").endswith("hex")
        n = int(w.replace("adpm", "").replace("_shipped", "").replace("hex", "") or 60)
        conn, xyz = synth.hex_mesh(n, jitter=0.1) if hexm else synth.kuhn_tet_mesh(n)
        u, tracts = synth.adpm_fields(xyz, conn.shape[0])
        shipped = w.endswith("_shipped")
        p = adpm_params_from_dict(synth.adpm_param_dict("shipped" if shipped else "full"), time=3.0)
        def setup(c):
            c.field_upload(FIELD_OLD_SOLUTION, u); c.field_upload(FIELD_ELEM_FIBRE, tracts)
        run(f"ADPM {'HEX8 H' if hexm else 'TET4 K'}({n})" + (", shipped run/HCP102513 parameters (decay only)" if shipped else ""), 8 if hexm else 4, conn, xyz, 3, setup, lambda c: c.assemble_adpm(p), 2, reps=6)
    elif w == "hcc126_shipped":   # run/Coupled/HCC/input.dat: every rate zero -> HccMassOnly
        conn, xyz = synth.hex_mesh(126, jitter=0.1)
        p, u = hcc_params_from_dict(synth.hcc_param_dict("shipped")), synth.hcc_fields(xyz)
        run("HCC HEX8 H(126), shipped run/Coupled/HCC parameters (all rates zero)", 8, conn, xyz, 3, lambda c: c.field_upload(FIELD_OLD_SOLUTION, u), lambda c: c.assemble_hcc(p), 2, reps=6)
    elif w == "hcc126":
        conn, xyz = synth.hex_mesh(126, jitter=0.1)
        p, u = hcc_params_from_dict(synth.hcc_param_dict("full")), synth.hcc_fields(xyz)
        for sc in (2, 1):
            run("HCC HEX8 H(126)", 8, conn, xyz, 3, lambda c: c.field_upload(FIELD_OLD_SOLUTION, u), lambda c: c.assemble_hcc(p), sc, reps=6)
    elif w.startswith("solid"):
        n = int(w.replace("solid", "").replace("tet", "") or 63)
        tet = w.endswith("tet")
        conn, Xu = synth.kuhn_tet_mesh(n, jitter=0.1) if tet else synth.hex_mesh(n, jitter=0.1)
        nen = 4 if tet else 8
        x = Xu + synth.solid_displacement(Xu)
        em = (np.linalg.norm(Xu[conn].mean(axis=1) - 0.5, axis=1) < 0.3).astype(np.int32)
        mats = [SolidMaterial(2.0e3, 0.4, 0.0, (0.0, 0.0, 0.0)), SolidMaterial(2.0e3, 0.4, 0.0, (0.3, 0.3, 0.3))]
        se0, ss0 = synth.boundary_sides(nen, conn, Xu, 2, 0.0)
        sd = np.zeros((se0.size, 3))
        sp = SolidParams(0.4, 1.0e8, 0, 0)
        def setup(c):
            c.field_upload(FIELD_UNDEFORMED_XYZ, Xu); c.field_upload(FIELD_ELEM_FIBRE, np.tile([0.0, 0.0, 1.0], (conn.shape[0], 1)))
            c.solid_set_materials(em, mats); c.solid_set_sides(se0, ss0, sd)
        # solid_kernel: 0 = default (fused cluster kernel for HEX8 tangents, else two-pass), 2 = two-pass, 1 = coloured
        for sk, sg, ss in (((0, 0, 0), (2, 0, 0), (2, 1, 0), (2, 0, 1), (1, 0, 0)) if conn.shape[0] <= 300000 else ((0, 0, 0), (2, 0, 0))):
            o = (("solid_kernel", sk), ("solid_gather", sg), ("solid_split", ss))
            run(f"SOLID {'TET4 K' if tet else 'HEX8 H'}({n}) residual+Jacobian, solid_kernel={sk} gather={sg} split={ss}", nen, conn, x, 3, setup,
                lambda c: c.solid_assemble(sp, True), 1, reps=6, n_in=0, solid=True, opts=o)
            if sg == 0 and ss == 0:
                run(f"SOLID {'TET4 K' if tet else 'HEX8 H'}({n}) residual only, solid_kernel={sk}", nen, conn, x, 3, setup,
                    lambda c: c.solid_assemble(sp, False), 1, reps=6, n_in=0, solid=True, opts=o)
