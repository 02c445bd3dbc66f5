"""Synthetic meshes, fields and parameter sets — the large meshes of the reference's
examples are not in its checkout (.MISSING_LARGE_BLOBS), so every benchmark / parity input is built
here, seeded and reproducible.

  K(n): unit cube, n^3 cells, 6 Kuhn tets per cell (positively oriented for libMesh's TET4 map)
  H(n): unit cube, n^3 HEX8 cells
"""
from __future__ import annotations

import itertools

import numpy as np

SEED = 20241016


def _grid_nodes(n, jitter, rng):
    g = np.arange(n + 1, dtype=np.float64) / n
    z, y, x = np.meshgrid(g, g, g, indexing="ij")  # node id = (k*(n+1) + j)*(n+1) + i, x fastest
    xyz = np.stack([x.ravel(), y.ravel(), z.ravel()], axis=1)
    if jitter > 0:
        h = 1.0 / n
        interior = np.all((xyz > 1e-12) & (xyz < 1 - 1e-12), axis=1)
        d = rng.uniform(-jitter * h, jitter * h, size=xyz.shape)
        xyz[interior] += d[interior]
    return xyz


def _cell_corner_ids(n):
    c = np.arange(n, dtype=np.int64)
    k, j, i = np.meshgrid(c, c, c, indexing="ij")
    base = ((k * (n + 1) + j) * (n + 1) + i).ravel()

    def corner(dx, dy, dz):
        return base + (dz * (n + 1) + dy) * (n + 1) + dx

    return corner


def kuhn_tet_mesh(n, jitter=0.2, seed=SEED, order="lex"):
    """K(n): returns (conn uint32 [6n^3][4], xyz float64 [(n+1)^3][3])."""
    rng = np.random.default_rng(seed)
    xyz = _grid_nodes(n, jitter, rng)
    corner = _cell_corner_ids(n)
    tets = []
    for perm in itertools.permutations(range(3)):
        v = [np.zeros(3, dtype=int)]
        for ax in perm:
            w = v[-1].copy()
            w[ax] = 1
            v.append(w)
        ids = [corner(*p) for p in v]
        # parity of the permutation = orientation of (e_a, e_b, e_c)
        inv = sum(1 for a in range(3) for b in range(a + 1, 3) if perm[a] > perm[b])
        if inv % 2 == 1:
            ids[2], ids[3] = ids[3], ids[2]
        tets.append(np.stack(ids, axis=1))
    conn = np.stack(tets, axis=1).reshape(-1, 4)  # the 6 tets of a cell are adjacent
    return _reorder(conn, xyz, order, rng)


def hex_mesh(n, jitter=0.0, seed=SEED, order="lex"):
    """H(n): returns (conn uint32 [n^3][8], xyz) in libMesh/Gmsh HEX8 node order."""
    rng = np.random.default_rng(seed)
    xyz = _grid_nodes(n, jitter, rng)
    corner = _cell_corner_ids(n)
    order8 = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
    conn = np.stack([corner(*p) for p in order8], axis=1)
    return _reorder(conn, xyz, order, rng)


def _reorder(conn, xyz, order, rng):
    if order == "lex":
        pass
    elif order == "random":
        pn = rng.permutation(xyz.shape[0])  # new id of old node
        inv = np.empty_like(pn)
        inv[pn] = np.arange(pn.size)
        xyz = xyz[inv]
        conn = pn[conn]
        conn = conn[rng.permutation(conn.shape[0])]
    else:
        raise ValueError(f"unknown order {order!r}")
    return np.ascontiguousarray(conn, dtype=np.uint32), np.ascontiguousarray(xyz)


# ---------------------------------------------------------------------------------------------
# parameter sets (reference string keys)
# ---------------------------------------------------------------------------------------------
def pihna_param_dict(variant="shipped"):
    """run/PIHNA/input.dat:9-49; 'full' turns on every transport term so no block is trivially 0."""
    d = {
        "time_step": 0.1, "cells_min_capacity": 1.0, "cells_max_capacity": 2.39e5,
        "cells_max_capacity/exponent": 3.0, "cytokines_max_capacity": 1.0e-8,
        "necrosis/c": 500.0, "necrosis/h": 200.0, "necrosis/v": 300.0,
        "diffuse/c": 0.0, "taxis/c": 0.0, "diffuse/h": 0.0, "taxis/h": 0.0, "produce/c": -2.5,
        "switch/c/to/h": 1.0, "switch/h/to/c": 1.82, "switch/h/to/n": 0.5,
        "diffuse/v": 0.5, "taxis/v": 0.0, "produce/v": 10.0,
        "secrete/a/from/c": 2.77e-13, "secrete/a/from/h": 5.22e-10, "uptake/a/from/v": 0.0, "decay/a": 5678.4,
    }
    if variant == "full":
        d.update({"diffuse/c": 0.1, "taxis/c": 0.1, "diffuse/h": 0.1, "taxis/h": 0.1, "taxis/v": 0.1,
                  "uptake/a/from/v": 1.0e-5})
    elif variant == "realexp":  # non-integer crowding exponent -> general pow() path
        d.update({"cells_max_capacity/exponent": 2.5, "diffuse/c": 0.1, "taxis/c": 0.1, "diffuse/h": 0.1,
                  "taxis/h": 0.1, "taxis/v": 0.1, "uptake/a/from/v": 1.0e-5})
    elif variant != "shipped":
        raise ValueError(variant)
    return d


def ripf_param_dict(variant="shipped"):
    """run/RIPF133/input.dat:8-36 (+ runtime RT_dose/total/max = 73)."""
    d = {
        "time_step": 0.1, "volume_fraction/stroma": 0.30, "volume_fraction/parenchyma": 0.20,
        "volume_fraction/exponent": 2.5, "volume_fraction/min_vacant": 1.0e-5,
        "HU/phi/tolerance": 1.0e-3, "cc/delta": 0.0864, "cc/delta/RT/a": 0.3, "cc/delta/RT/b": 0.03,
        "fb/lambda": 0.01, "fb/lambda/RT/r": 1.0, "fb/omega": 0.1, "fb/diffusion": 1.0e-20,
        "fb/haptotaxis": 0.05, "RT_dose/total/max": 73,
    }
    if variant == "full":
        d.update({"HU/phi/cc/build": 0.3, "HU/phi/cc/decay": -0.2, "HU/phi/cc/rate": 0.05, "HU/phi/fb/build": 0.4,
                  "HU/phi/fb/decay": -0.1, "HU/phi/fb/rate": 0.07, "cc/kappa": 0.01, "cc/kappa/RT/c": 0.02,
                  "fb/lambda/RT/r": 0.0, "fb/lambda/HU/r": -600.0, "fb/omicro": 0.02, "fb/omicro/RT/r": 0.0,
                  "fb/omicro/fb/b": 0.05, "fb/diffusion": 1.0e-3, "fb/radiotaxis": 0.03,
                  "volume_fraction/exponent": 3.0})
    elif variant != "shipped":
        raise ValueError(variant)
    return d


def hcc_param_dict(variant="full"):
    """run/Coupled/HCC/input.dat gives only the capacity keys (all rates default to 0); 'full' adds
    non-zero rates so every touched block is exercised."""
    d = {"time_step": 0.01, "cells/min_capacity": 0.0, "cells/max_capacity": 1.0, "cells/max_capacity/exponent": 3.0}
    if variant == "full":
        d.update({"produce/l": 0.8, "diffuse/c": 0.05, "mechano/c": 0.02, "produce/c": 1.2, "necrosis/l": 0.3,
                  "necrosis/c": 0.4, "necrosis/pressure": 0.1})
    elif variant != "shipped":
        raise ValueError(variant)
    return d


def adpm_param_dict(variant="shipped"):
    """run/HCP102513/input.dat as the code reads it: its `taxis/A_b*`, `taxis/Tau*` keys are not the
    `taxis_1/...`, `taxis_2/...` keys of input() (src/adpm.C:194-199,216-221), so the shipped run has no taxis;
    'full' switches every term on with thresholds inside the range of adpm_fields()."""
    d = {"time_step": 0.05, "decay/PrP": 1.0e-4, "decay/PrP/pulse/0": 0.01, "decay/PrP/pulse/1": 10.0,
         "decay/Tau": 10.0, "decay/Tau/pulse/0": 0.0005}
    if variant == "full":
        d.update({"decay/PrP/time_exponent": 0.5,
                  "transform/A_b": 0.3, "transform/A_b/trapezoid/0": 0.001, "transform/A_b/trapezoid/1": 0.006,
                  "transform/A_b/trapezoid/2": 0.012, "transform/A_b/trapezoid/3": 0.018,
                  "transform/Tau": 0.2, "transform/Tau/trapezoid/0": 0.002, "transform/Tau/trapezoid/1": 0.005,
                  "transform/Tau/trapezoid/2": 0.010, "transform/Tau/trapezoid/3": 0.019,
                  "diffuse/A_b": 0.02, "diffuse/A_b/pulse/0": 0.002, "diffuse/A_b/pulse/1": 0.5,
                  "taxis/A_b/angle": 60.0, "taxis_1/A_b": 0.4, "taxis_1/A_b/pulse/0": 0.001, "taxis_1/A_b/pulse/1": 0.015,
                  "taxis_2/A_b": 0.1, "taxis_2/A_b/pulse/0": 0.004, "taxis_2/A_b/pulse/1": 0.5,
                  "produce/A_b": 0.7, "produce/A_b/sigmoid/0": 0.005, "produce/A_b/sigmoid/1": 0.015,
                  "decay/A_b": 0.5, "decay/A_b/pulse/0": 0.008, "decay/A_b/pulse/1": 0.5,
                  "diffuse/Tau": 0.03, "diffuse/Tau/pulse/0": 0.001, "diffuse/Tau/pulse/1": 0.5,
                  "taxis/Tau/angle": 45.0, "taxis_1/Tau": 0.3, "taxis_1/Tau/pulse/0": 0.002, "taxis_1/Tau/pulse/1": 0.5,
                  "taxis_2/Tau": 0.2, "taxis_2/Tau/pulse/0": 0.003, "taxis_2/Tau/pulse/1": 0.012,
                  "produce/Tau": 0.6, "produce/Tau/sigmoid/0": 0.004, "produce/Tau/sigmoid/1": 0.016})
    elif variant != "shipped":
        raise ValueError(variant)
    return d


def proteas_param_dict(variant="full"):
    """No PROTEAS example ships with the reference; 'defaults' = input()'s all-ones (src/proteas.C:180-212),
    'full' = distinct values with the thresholds inside the range of proteas_fields() and a real RT exponent."""
    if variant == "defaults":
        return {"time_step": 0.05}
    if variant != "full":
        raise ValueError(variant)
    return {"time_step": 0.05, "cells/total_capacity": 1.6, "radiotherapy/max_dosage": 60.0,
            "host/proliferation": 0.3, "host/vsc_threshold": 0.05, "host/RT_death_rate": 0.2, "host/RT_exp_a": 0.03,
            "host/RT_exp_b": 0.002, "host/necrosis_rate": 0.15, "tumour/diffusion": 0.02, "tumour/diffusion_host": 0.01,
            "tumour/proliferation": 0.8, "tumour/vsc_threshold": 0.08, "tumour/RT_death_rate": 0.5, "tumour/RT_exp_a": 0.05,
            "tumour/RT_exp_b": 0.004, "tumour/necrosis_rate": 0.25, "necrosis/clearance": 0.1, "necrosis/slope": 6.0,
            "necrosis/vsc_threshold": 0.5, "vascular/proliferation": 0.4, "vascular/necrosis_rate": 0.2,
            "oedema/diffusion": 0.05, "oedema/proliferation": 0.6, "oedema/vsc_threshold": 0.1, "oedema/RT_coeff": 0.3,
            "oedema/RT_exp": 1.5, "oedema/reabsorption_rate": 0.35}


# ---------------------------------------------------------------------------------------------
# fields
# ---------------------------------------------------------------------------------------------
def proteas_fields(xyz, seed=SEED):
    """([n_node][5] (hos, tum, nec, vsc, oed) volume fractions, [n_node][3] aux = {HU, RTD, 0}); the dose-like
    aux component 0 is positive (upstream raises RTD/RT_max to a real power)."""
    rng = np.random.default_rng(seed + 11)
    n = xyz.shape[0]
    r = np.linalg.norm(xyz - 0.5, axis=1)
    u = np.empty((n, 5))
    u[:, 0] = 0.5 + 0.2 * np.cos(3.0 * xyz[:, 0]) + rng.uniform(-0.02, 0.02, n)
    u[:, 1] = 0.35 * np.exp(-(r / 0.3) ** 2) + rng.uniform(0.0, 0.02, n)
    u[:, 2] = 0.1 * np.exp(-(r / 0.15) ** 2) + rng.uniform(0.0, 0.01, n)
    u[:, 3] = 0.12 + 0.08 * np.sin(4.0 * xyz[:, 1]) + rng.uniform(-0.01, 0.01, n)
    u[:, 4] = 0.2 * np.exp(-(r / 0.4) ** 2) + rng.uniform(0.0, 0.02, n)
    aux = np.zeros((n, 3))
    aux[:, 0] = 5.0 + 40.0 * np.exp(-(r / 0.35) ** 2) + rng.uniform(0.0, 1.0, n)
    aux[:, 1] = rng.uniform(0.0, 60.0, n)   # never read by the assembly (src/proteas.C:481 indexes variable 0)
    return u, aux


def adpm_fields(xyz, n_elem, seed=SEED):
    """([n_node][3] (PrP, A_b, Tau), [n_elem][3] tract vectors): PrP around the shipped background 1
    (run/HCP102513/Brain_Model_Initial_Nodal_Field.dat), misfolded species in [0, 0.02] with smooth parts so
    that their gradients have a direction; tracts with the magnitude of the shipped elemental file (~0.1)."""
    rng = np.random.default_rng(seed + 7)
    n = xyz.shape[0]
    u = np.empty((n, 3))
    u[:, 0] = rng.uniform(0.5, 1.5, n)
    u[:, 1] = 0.01 + 0.008 * np.sin(5.0 * xyz[:, 0] + 1.0) * np.cos(3.0 * xyz[:, 1]) + rng.uniform(-0.002, 0.002, n)
    u[:, 2] = 0.01 + 0.008 * np.cos(4.0 * xyz[:, 2] + 0.5) * np.sin(6.0 * xyz[:, 0]) + rng.uniform(-0.002, 0.002, n)
    tracts = 0.1 * rng.standard_normal((n_elem, 3))
    return u, tracts



def pihna_fields(xyz, seed=SEED, radius=0.25):
    """[n_node][5] (n,c,h,v,a): background (0,0,0,7170,0) as run/PIHNA/Brain_Model_Initial_Nodal_Field.dat,
    a non-degenerate tumour state inside a sphere r=0.25 (v > 0 everywhere: the 0/0 path of
    src/pihna.C:477 is exercised separately).  radius > 0.9: the tumour state at every node ("dense" state:
    no element is in the background state the element-visit kernel takes its short cut for)."""
    rng = np.random.default_rng(seed + 1)
    n = xyz.shape[0]
    u = np.zeros((n, 5))
    u[:, 3] = 7170.0
    r = np.linalg.norm(xyz - 0.5, axis=1)
    s = r < radius
    k = int(s.sum())
    u[s, 0] = rng.uniform(0, 5e2, k)
    u[s, 1] = rng.uniform(0, 2e3, k)
    u[s, 2] = rng.uniform(0, 2e3, k)
    u[s, 3] = rng.uniform(3e3, 7.17e3, k)
    u[s, 4] = rng.uniform(0, 1e-8, k)
    return u


def ripf_fields(xyz, seed=SEED):
    """returns (u [n][3] = HU,cc,fb ; aux [n][3] = cc_dtime, fb_dtime, RT_total), mimicking
    run/RIPF133/*.dat value ranges."""
    rng = np.random.default_rng(seed + 2)
    n = xyz.shape[0]
    u = np.zeros((n, 3))
    u[:, 0] = rng.uniform(-1019.0, 1094.0, n)
    r2 = np.sum((xyz - np.array([0.45, 0.55, 0.5])) ** 2, axis=1)
    s = r2 < 0.3 ** 2
    k = int(s.sum())
    u[s, 1] = rng.uniform(0, 1, k)
    u[s, 2] = rng.uniform(0, 0.3, k)
    aux = np.empty((n, 3))
    aux[:, 0] = rng.uniform(-1e-2, 1e-2, n)
    aux[:, 1] = rng.uniform(-1e-2, 1e-2, n)
    aux[:, 2] = 67.0 * np.exp(-r2 / 0.1) + 6.7 * np.exp(-r2 / 0.01)
    return u, aux


def hcc_fields(xyz, seed=SEED):
    rng = np.random.default_rng(seed + 3)
    return rng.uniform(0.0, 0.3, (xyz.shape[0], 3))


def solid_displacement(xyz, amp=0.02):
    """smooth displacement field used to deform the mesh for cfg5-style tests."""
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    d = np.stack([np.sin(2 * np.pi * y) * np.cos(np.pi * z), np.sin(2 * np.pi * z) * np.cos(np.pi * x),
                  np.sin(2 * np.pi * x) * np.cos(np.pi * y)], axis=1)
    return amp * d


def boundary_sides(elem_type, conn, xyz, axis, value, tol=1e-9):
    """(elem, side) pairs whose side nodes all lie on the plane xyz[:,axis] == value.
    Side numbering = libMesh Tet4/Hex8 side_nodes_map."""
    if elem_type == 4:
        sides = [(0, 2, 1), (0, 1, 3), (1, 2, 3), (2, 0, 3)]
    else:
        sides = [(0, 3, 2, 1), (0, 1, 5, 4), (1, 2, 6, 5), (2, 3, 7, 6), (3, 0, 4, 7), (4, 5, 6, 7)]
    on = np.abs(xyz[:, axis] - value) < tol
    es, ss = [], []
    for s, nodes in enumerate(sides):
        m = np.all(on[conn[:, nodes]], axis=1)
        idx = np.nonzero(m)[0]
        es.append(idx)
        ss.append(np.full(idx.size, s, dtype=np.int32))
    return np.concatenate(es).astype(np.int64), np.concatenate(ss)
