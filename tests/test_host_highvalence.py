"""High-valence unstructured TET4 mesh (Delaunay of random points: max valence ~35-40) through the host list builders.
The row-gather schedule (rdc_meshprep.cpp) and the element-visit lists (rdc_prep_ev.cpp) have fixed per-workgroup limits
(16 node blocks per row in the moment slice, 64 lanes x private copies per node and wave): such a mesh must be rejected
GRACEFULLY -- rg2_ok = 0 / an error string, nothing written past the lists -- so that rdc_mesh_upload falls back to the
generic kernels.  Run once under AddressSanitizer with tools/asan_prep.sh (heap-buffer-overflow at
rdc_prep_ev.cpp:234 and rdc_meshprep.cpp:434-448)."""
import ctypes as C

import numpy as np
import pytest


def delaunay_tets(n_points=3000, seed=7):
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(seed)
    xyz = rng.random((n_points, 3))
    tets = Delaunay(xyz).simplices.astype(np.uint32)
    # drop slivers (degenerate volume) so the mesh is a legal FE mesh
    v = np.einsum("ij,ij->i", np.cross(xyz[tets[:, 1]] - xyz[tets[:, 0]], xyz[tets[:, 2]] - xyz[tets[:, 0]]), xyz[tets[:, 3]] - xyz[tets[:, 0]])
    tets = tets[np.abs(v) > 1e-9]
    return np.ascontiguousarray(tets), xyz


def test_high_valence_mesh_is_rejected_gracefully(shim, make_prep):
    conn, xyz = delaunay_tets()
    nn = xyz.shape[0]
    deg = np.zeros(nn, dtype=np.int64)
    pairs = set()
    for a in range(4):
        for b in range(4):
            if a != b:
                pairs.update(zip(conn[:, a].tolist(), conn[:, b].tolist()))
    for i, _ in pairs:
        deg[i] += 1
    assert deg.max() > 15                                        # beyond the 16 node blocks per row of the moment slice
    for conflict_aware_nvar in (5, 3):
        P = make_prep(4, conn, nn, nn, conflict_aware_nvar)
        assert P.ok, P.error                                      # pattern, slots, colouring are fine
        assert P.bptr[-1] == len(pairs) + nn
        stats = (C.c_int64 * 6)()
        rc = shim.shim_ev_build(C.c_int64(54000), stats)
        assert rc == 1
        assert b"16 node blocks" in shim.shim_prep_error()
    # the staged row gather either fits or is switched off, never half-written: its lists stay inside their arrays
    if P.rg2_ok:
        assert P.pair_rec.size == P.wg2.size * P.rg2_block * 4


@pytest.mark.parametrize("nvar", [5, 3])
def test_kuhn_mesh_still_builds(shim, make_prep, nvar):
    from rdcfes_amd import synth
    conn, xyz = synth.kuhn_tet_mesh(6, order="random")
    P = make_prep(4, conn, xyz.shape[0], xyz.shape[0], nvar)
    assert P.ok and P.rg2_ok
    stats = (C.c_int64 * 6)()
    assert shim.shim_ev_build(C.c_int64(54000), stats) == 0, shim.shim_prep_error()
    assert stats[5] == xyz.shape[0]
