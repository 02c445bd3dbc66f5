import sys, numpy as np
sys.path.insert(0, '.')
from rdcfes_amd import AssemblyContext, pihna_params_from_dict, synth
from rdcfes_amd.context import FIELD_OLD_SOLUTION
conn, xyz = synth.kuhn_tet_mesh(119)
p = pihna_params_from_dict(synth.pihna_param_dict("shipped"))
pf = pihna_params_from_dict(synth.pihna_param_dict("full"))
u = synth.pihna_fields(xyz); ud = synth.pihna_fields(xyz, radius=10.0)
with AssemblyContext(0) as ctx:
    ctx.mesh_upload(4, conn, xyz, 5)
    ref = {}
    worst = 0.0
    for it in range(120):
        key = ("sparse" if it % 2 == 0 else "dense", "shipped" if (it // 2) % 2 == 0 else "full")
        ctx.field_upload(FIELD_OLD_SOLUTION, u if key[0] == "sparse" else ud)
        ctx.assemble_pihna(p if key[1] == "shipped" else pf)
        if it % 8 < 4 or it > 110:
            val, rhs = ctx.csr_download()
            s = (float(np.abs(val).sum()), float(np.abs(rhs).sum()))
            if key not in ref: ref[key] = (val.copy(), rhs.copy())
            else:
                d = max(np.abs(val - ref[key][0]).max() / np.abs(ref[key][0]).max(), np.abs(rhs - ref[key][1]).max() / np.abs(ref[key][1]).max())
                worst = max(worst, d)
    print("soak: 120 assemblies, 4 state/parameter combinations, worst relative difference between repeats", worst, "finite", np.isfinite(val).all())
