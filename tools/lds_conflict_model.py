#!/usr/bin/env python3
"""Host-side estimate of the LDS FP64-atomic pass count of a row-gather schedule, with the bank model
measured by tools/lds_bank_model.hip on MI355X: a ds_add_f64 wave instruction is executed in four groups
of 16 consecutive lanes; within a group, lanes whose double-index is equal modulo 16 are serialised
(1.56 CU-cycles per pass).  Prints the mean passes per 16-lane group (1.0 = conflict-free)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
import conftest as CT
from rdcfes_amd import synth


def passes(P, nvar=5, block=256, ncop=6):
    NW = block // 64
    nwg = P.wg2.shape[0]
    ax = P.pair_aux.reshape(nwg, block, 8).astype(np.int64)
    valid = P.pair_rec.reshape(nwg, block, 4)[:, :, 0] != 0xFFFFFFFF
    idx = np.arange(block)
    wave, lane = idx % NW, idx // NW
    grp = wave * 4 + lane // 16                                   # 16 groups of 16 lanes
    tot = cnt = 0
    dtot = dcnt = 0
    for g in range(NW * 4):
        m = grp == g
        v = valid[:, m]
        for j in range(1, 4):
            for a in range(nvar):
                bank = (ax[:, m, 0] + a * ax[:, m, 1] + ax[:, m, 4 + j]) & 15
                occ = np.zeros((nwg, 16), np.int64)
                for l in range(bank.shape[1]):
                    np.add.at(occ, (np.arange(nwg)[v[:, l]], bank[v[:, l], l]), 1)
                mx = occ.max(1)
                tot += mx[mx > 0].sum(); cnt += (mx > 0).sum()
        slot = (ax[:, m, 2] // nvar) * ncop + ax[:, m, 3]
        occ = np.zeros((nwg, 16), np.int64)
        for l in range(slot.shape[1]):
            np.add.at(occ, (np.arange(nwg)[v[:, l]], slot[v[:, l], l] & 15), 1)
        mx = occ.max(1)
        dtot += mx[mx > 0].sum(); dcnt += (mx > 0).sum()
    return tot / cnt, dtot / dcnt


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    order = sys.argv[2] if len(sys.argv) > 2 else "lex"
    lib = CT.build_shim()
    conn, xyz = synth.kuhn_tet_mesh(n, order=order)
    t = time.time()
    P = CT.Prep(lib, 4, conn, xyz.shape[0], xyz.shape[0], 5, lds_budget=50 * 1024)
    dt = time.time() - t
    off, diag = passes(P)
    print(f"K({n}) {order}: prep {dt:.2f} s, WGs {P.wg2.shape[0]}, off-diagonal passes/group {off:.3f}, diagonal passes/group {diag:.3f}")
