import ctypes as C
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


def build_shim():
    """g++ build of the product's host/device headers (tests/host_shim.cpp)."""
    if os.environ.get("RDC_SHIM_SO"):      # e.g. the AddressSanitizer build of tools/asan_prep.sh
        lib = C.CDLL(str(ROOT / os.environ["RDC_SHIM_SO"]))
        lib.shim_prep_size.restype = C.c_int64
        lib.shim_prep_size.argtypes = [C.c_int]
        lib.shim_prep_error.restype = C.c_char_p
        return lib
    bdir = ROOT / "tests" / "_build"
    bdir.mkdir(exist_ok=True)
    so = bdir / "libhost_shim.so"
    srcs = [ROOT / "tests" / "host_shim.cpp", ROOT / "rdcfes_amd" / "csrc" / "rdc_meshprep.cpp",
            ROOT / "rdcfes_amd" / "csrc" / "rdc_prep_ev.cpp", ROOT / "rdcfes_amd" / "csrc" / "rdc_prep_cl.cpp"]
    deps = srcs + list((ROOT / "rdcfes_amd" / "csrc").glob("*.h")) + [ROOT / "include" / "rdc_assembly.h"]
    if not so.exists() or so.stat().st_mtime < max(p.stat().st_mtime for p in deps):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-Wno-unknown-pragmas",
               "-o", str(so)] + [str(s) for s in srcs]
        subprocess.run(cmd, check=True)
    lib = C.CDLL(str(so))
    lib.shim_prep_size.restype = C.c_int64
    lib.shim_prep_size.argtypes = [C.c_int]
    lib.shim_prep_error.restype = C.c_char_p
    return lib


@pytest.fixture(scope="session")
def shim():
    return build_shim()


class Prep:
    """numpy view of rdc::HostPrep built through the shim."""
    NAMES = {"bptr": (0, np.int64), "bcol": (1, np.int32), "eslot": (2, np.uint16), "colour": (3, np.int32),
             "elem_order": (4, np.uint32), "colour_ptr": (5, np.int64), "first_mask": (6, np.uint64),
             "first_rhs": (7, np.uint8), "pair_elem": (8, np.uint32), "pair_local": (9, np.uint8),
             "node_pair_ptr": (10, np.int64), "wg_node_ptr": (11, np.int32)}

    def __init__(self, lib, nen, conn, n_node, n_owned, nvar, lds_budget=60 * 1024, block=256):
        conn = np.ascontiguousarray(conn, dtype=np.uint32)
        rc = lib.shim_prep_build(nen, C.c_int64(conn.shape[0]), C.c_int64(n_node), C.c_int64(n_owned),
                                 conn.ctypes.data_as(C.POINTER(C.c_uint32)), nvar, C.c_int64(lds_budget), block)
        self.ok = rc == 0
        self.error = lib.shim_prep_error().decode()
        if not self.ok:
            return
        for name, (idx, dt) in self.NAMES.items():
            a = np.empty(lib.shim_prep_size(idx), dtype=dt)
            if a.size:
                lib.shim_prep_copy(idx, a.ctypes.data_as(C.c_void_p))
            setattr(self, name, a)
        WG = np.dtype([("vb0", "<i8"), ("bb0", "<i8"), ("c0", "<i8"), ("ch0", "<i8"), ("n0", "<i4"), ("nnodes", "<i4"),
                       ("nb", "<i4"), ("np", "<i4"), ("nch", "<i4"), ("nout", "<i4"), ("pad", "<i8")])
        CH = np.dtype([("cbeg", "<u2"), ("cnt", "<u2"), ("dst", "<u2"), ("pad", "<u2")])
        SD = np.dtype([("outoff", "<u2"), ("len", "u1"), ("nextra", "u1"), ("extra", "<u2"), ("diag", "u1"), ("node", "u1")])
        for name, idx, dt in (("wg2", 12, WG), ("pair_rec", 13, np.uint32), ("pair_aux", 14, np.uint16),
                              ("chunk", 15, CH), ("sdesc", 16, SD), ("contrib", 17, np.uint16)):
            nbytes_or_count = lib.shim_prep_size(idx)
            count = nbytes_or_count // np.dtype(dt).itemsize if idx in (12, 15, 16) else nbytes_or_count
            a = np.empty(count, dtype=dt)
            if a.size:
                lib.shim_prep_copy(idx, a.ctypes.data_as(C.c_void_p))
            setattr(self, name, a)
        self.rg2_ok = bool(lib.shim_prep_size(103))
        self.rg2_lds_bytes = lib.shim_prep_size(104)
        self.rg2_block = lib.shim_prep_size(105)
        self.n_colours = lib.shim_prep_size(100)
        self.rowgather_ok = bool(lib.shim_prep_size(101))
        self.rg_lds_bytes = lib.shim_prep_size(102)


@pytest.fixture(scope="session")
def make_prep(shim):
    def f(nen, conn, n_node, n_owned, nvar, **kw):
        return Prep(shim, nen, conn, n_node, n_owned, nvar, **kw)
    return f


def shim_rows(lib, model, nen, params, X, U, A=None, fast=False, force_general_pow=False, elem_data=None):
    """all rows through the product's row function -> (Ke [nv*nen][nv*nen] var-major, Fe)"""
    nv = {0: 5, 1: 3, 2: 3, 3: 5, 4: 3, 5: 5, 6: 5, 7: 3, 8: 3, 9: 3}[model]
    ED = None if elem_data is None else np.ascontiguousarray(elem_data, dtype=np.float64)
    X = np.ascontiguousarray(X, dtype=np.float64)
    U = np.ascontiguousarray(U, dtype=np.float64)
    A = None if A is None else np.ascontiguousarray(A, dtype=np.float64)
    Ke = np.zeros((nv * nen, nv * nen))
    Fe = np.zeros(nv * nen)
    dp = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))
    for i in range(nen):
        acc = np.empty((nv, nv, nen))
        fe = np.empty(nv)
        rc = lib.shim_row(model, nen, int(fast), int(force_general_pow), C.byref(params), dp(X), dp(U), dp(A), i,
                          dp(acc), dp(fe), dp(ED))
        assert rc == 0
        for a in range(nv):
            Fe[a * nen + i] = fe[a]
            for b in range(nv):
                Ke[a * nen + i, b * nen:(b + 1) * nen] = acc[a, b]
    return Ke, Fe
