#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_regression.npz from the CPU oracle (drift guard, not a reference pin)."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import oracle as O
from rdcfes_amd import (SolidMaterial, SolidParams, adpm_params_from_dict, hcc_params_from_dict, pihna_params_from_dict,
                        proteas_params_from_dict, ripf_params_from_dict, synth)


def cases():
    out = {}
    conn, xyz = synth.kuhn_tet_mesh(2, order="random")
    hconn, hxyz = synth.hex_mesh(2, jitter=0.1, order="random")
    u = synth.pihna_fields(xyz)
    u[:, 1:3] += 50.0  # some tumour everywhere so the tiny mesh is non-degenerate
    out["pihna_tet4"] = O.assemble(0, 4, conn, xyz, 5, pihna_params_from_dict(synth.pihna_param_dict("full")), u_old=u)[2:]
    ur, aux = synth.ripf_fields(xyz)
    ur[:, 1] = 0.3; ur[:, 2] = 0.1
    out["ripf_tet4"] = O.assemble(1, 4, conn, xyz, 3, ripf_params_from_dict(synth.ripf_param_dict("full")), u_old=ur, aux=aux)[2:]
    out["hcc_hex8"] = O.assemble(2, 8, hconn, hxyz, 3, hcc_params_from_dict(synth.hcc_param_dict("full")), u_old=synth.hcc_fields(hxyz))[2:]
    x = hxyz + synth.solid_displacement(hxyz)
    mats = [SolidMaterial(2.0e3, 0.4, 30.0, (0.3, 0.2, 0.1))]
    se, ss = synth.boundary_sides(8, hconn, hxyz, 2, 0.0)
    sd = np.tile([0.0, np.nan, -0.1], (se.size, 1))
    out["solid_hex8"] = O.assemble(3, 8, hconn, x, 3, SolidParams(0.4, 1e5, 0, 0), xyz_undeformed=hxyz,
                                   elem_fibre=np.tile([1.0, 2.0, 3.0], (hconn.shape[0], 1)),
                                   elem_material=np.zeros(hconn.shape[0], np.int32), materials=mats, sides=(se, ss, sd))[2:]
    ua, tr = synth.adpm_fields(xyz, conn.shape[0])
    out["adpm_tet4"] = O.assemble(4, 4, conn, xyz, 3, adpm_params_from_dict(synth.adpm_param_dict("full"), time=3.0), u_old=ua, elem_fibre=tr)[2:]
    up, ax = synth.proteas_fields(hxyz)
    out["proteas_hex8"] = O.assemble(5, 8, hconn, hxyz, 5, proteas_params_from_dict(synth.proteas_param_dict("full")), u_old=up, aux=ax)[2:]
    return out


if __name__ == "__main__":
    flat = {}
    for k, (val, rhs) in cases().items():
        flat[k + "_val"], flat[k + "_rhs"] = val, rhs
    np.savez_compressed(Path(__file__).with_name("oracle_regression.npz"), **flat)
    print({k: v.shape for k, v in flat.items()})
