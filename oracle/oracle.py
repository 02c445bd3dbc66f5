"""ctypes front-end of the CPU oracle (oracle/rdc_oracle.c).  TEST INFRASTRUCTURE ONLY — imported
by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by rdcfes_amd.
PARITY UNPINNED (the reference holds no golden vectors; see rdc_oracle.c header)."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from rdcfes_amd.params import HccParams, PihnaParams, RipfParams, SolidMaterial, SolidParams

HERE = Path(__file__).resolve().parent
LIB = HERE / "librdc_oracle.so"
MODEL_PIHNA, MODEL_RIPF, MODEL_HCC, MODEL_SOLID, MODEL_ADPM, MODEL_PROTEAS = 0, 1, 2, 3, 4, 5
FAST_LIB = HERE / "_fast" / "librdc_oracle_fast.so"
_lib = None
_fast = None


def build(force=False):
    src = [HERE / "rdc_oracle.c", HERE / "rdc_oracle.h", HERE.parent / "include" / "rdc_assembly.h"]
    if force or not LIB.exists() or LIB.stat().st_mtime < max(p.stat().st_mtime for p in src):
        subprocess.run(["make", "-C", str(HERE), "-B", "librdc_oracle.so"], check=True, capture_output=True)
    return LIB


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(LIB))
        _lib.oracle_build_node_pattern.restype = C.c_int64
    return _lib


REF_SRC = Path("/root/reference/src/eig3.C")
REF_LIB = HERE / "_ref" / "libref_eig3.so"
_ref = None


def build_ref():
    """oracle/_ref/libref_eig3.so: the reference's own src/eig3.C compiled where it lies (oracle/Makefile) -- only where
    /root/reference exists (this container); the GPU box uses the prebuilt file or goes without."""
    if REF_SRC.exists():
        src = [HERE / "ref_eig3_wrap.cpp", REF_SRC]
        if not REF_LIB.exists() or REF_LIB.stat().st_mtime < max(p.stat().st_mtime for p in src):
            subprocess.run(["make", "-C", str(HERE), "-B", "_ref/libref_eig3.so"], check=True, capture_output=True)
    return REF_LIB if REF_LIB.exists() else None


def ref_lib():
    """the reference's eigen_decomposition behind C linkage, or None when it has not been built"""
    global _ref
    if _ref is None and build_ref() is not None:
        _ref = C.CDLL(str(REF_LIB))
    return _ref


def fast_lib():
    """The timing build (-O3 -march=native, oracle/Makefile) for bench.py's cpu_baseline: always rebuilt on the
    machine it runs on (never shipped: -march=native code is only valid where it was compiled)."""
    global _fast
    if _fast is None:
        subprocess.run(["make", "-C", str(HERE), "-B", "_fast/librdc_oracle_fast.so"], check=True, capture_output=True)
        _fast = C.CDLL(str(FAST_LIB))
    return _fast


def _p(a, t=C.c_double):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def nqp(elem_type):
    return lib().oracle_nqp(int(elem_type))


def fe_reinit(elem_type, X):
    X = np.ascontiguousarray(X, dtype=np.float64)
    q = nqp(elem_type)
    phi = np.empty((q, elem_type))
    dphi = np.empty((q, elem_type, 3))
    jxw = np.empty(q)
    rc = lib().oracle_fe_reinit(int(elem_type), _p(X), _p(phi), _p(dphi), _p(jxw))
    assert rc == 0
    return phi, dphi, jxw


def element(model, elem_type, X, u, params, aux=None, elem_data=None):
    """Ke [nv*nen][nv*nen] (var-major), Fe for one element of an RD model (elem_data: ADPM tract vector)."""
    phi, dphi, jxw = fe_reinit(elem_type, X)
    nv = {MODEL_PIHNA: 5, MODEL_RIPF: 3, MODEL_HCC: 3, MODEL_ADPM: 3, MODEL_PROTEAS: 5}[model]
    nd = nv * elem_type
    Ke, Fe = np.empty((nd, nd)), np.empty(nd)
    u = np.ascontiguousarray(u, dtype=np.float64)
    L, q = lib(), nqp(elem_type)
    if model == MODEL_PIHNA:
        L.oracle_pihna_element(elem_type, q, _p(phi), _p(dphi), _p(jxw), _p(u), C.byref(params), _p(Ke), _p(Fe))
    elif model == MODEL_RIPF:
        aux = np.ascontiguousarray(aux, dtype=np.float64)
        L.oracle_ripf_element(elem_type, q, _p(phi), _p(dphi), _p(jxw), _p(u), _p(aux), C.byref(params), _p(Ke), _p(Fe))
    elif model == MODEL_PROTEAS:
        a0 = np.ascontiguousarray(np.asarray(aux, dtype=np.float64).reshape(elem_type, -1)[:, 0])
        L.oracle_proteas_element(elem_type, q, _p(phi), _p(dphi), _p(jxw), _p(u), _p(a0), C.byref(params), _p(Ke), _p(Fe))
    elif model == MODEL_ADPM:
        ed = np.ascontiguousarray(elem_data, dtype=np.float64)
        L.oracle_adpm_element(elem_type, q, _p(phi), _p(dphi), _p(jxw), _p(u), _p(ed), C.byref(params), _p(Ke), _p(Fe))
    else:
        L.oracle_hcc_element(elem_type, q, _p(phi), _p(dphi), _p(jxw), _p(u), C.byref(params), _p(Ke), _p(Fe))
    return Ke, Fe


def hyperelastic_point(gradX, lam, fibre, E, nu, K):
    gradX = np.ascontiguousarray(gradX, dtype=np.float64)
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    fibre = np.ascontiguousarray(fibre, dtype=np.float64)
    sig, Cm = np.empty((3, 3)), np.empty((6, 6))
    lib().oracle_hyperelastic_point(_p(gradX), _p(lam), _p(fibre), C.c_double(E), C.c_double(nu), C.c_double(K), _p(sig), _p(Cm))
    return sig, Cm


def solid_element(elem_type, x, Xu, fibre, material, pseudo_time, request_jacobian=True, use_symmetry=False):
    _, dphi, jxw = fe_reinit(elem_type, x)
    nd = 3 * elem_type
    Je, Re = np.empty((nd, nd)), np.empty(nd)
    Xu = np.ascontiguousarray(Xu, dtype=np.float64)
    fibre = np.ascontiguousarray(fibre, dtype=np.float64)
    lib().oracle_solid_element(elem_type, nqp(elem_type), _p(dphi), _p(jxw), _p(Xu), _p(fibre), C.byref(material),
                               C.c_double(pseudo_time), int(request_jacobian), int(use_symmetry), _p(Je), _p(Re))
    return Je, Re


def solid_side(elem_type, side, x, Xu, disp, pseudo_time, penalty, request_jacobian=True):
    nd = 3 * elem_type
    Je, Re = np.zeros((nd, nd)), np.zeros(nd)
    x = np.ascontiguousarray(x, dtype=np.float64)
    Xu = np.ascontiguousarray(Xu, dtype=np.float64)
    disp = np.ascontiguousarray(disp, dtype=np.float64)
    lib().oracle_solid_side(elem_type, int(side), _p(x), _p(Xu), _p(disp), C.c_double(pseudo_time), C.c_double(penalty),
                            int(request_jacobian), _p(Je), _p(Re))
    return Je, Re


def build_pattern(elem_type, conn, n_node, n_owned, nvar):
    """scalar CSR (row_ptr int64, col_idx int32) of the owned rows + node pattern (bptr, bcol)."""
    conn = np.ascontiguousarray(conn, dtype=np.uint32)
    L = lib()
    bptr = np.empty(n_owned + 1, dtype=np.int64)
    nb = L.oracle_build_node_pattern(elem_type, C.c_int64(conn.shape[0]), C.c_int64(n_node), C.c_int64(n_owned),
                                     _p(conn, C.c_uint32), _p(bptr, C.c_int64), None)
    bcol = np.empty(nb, dtype=np.int32)
    L.oracle_build_node_pattern(elem_type, C.c_int64(conn.shape[0]), C.c_int64(n_node), C.c_int64(n_owned),
                                _p(conn, C.c_uint32), _p(bptr, C.c_int64), _p(bcol, C.c_int32))
    row_ptr = np.empty(n_owned * nvar + 1, dtype=np.int64)
    col = np.empty(nb * nvar * nvar, dtype=np.int32)
    L.oracle_expand_pattern(nvar, C.c_int64(n_owned), _p(bptr, C.c_int64), _p(bcol, C.c_int32), _p(row_ptr, C.c_int64),
                            _p(col, C.c_int32))
    return row_ptr, col, bptr, bcol


def assemble(model, elem_type, conn, xyz, nvar, params, u_old=None, aux=None, n_owned=None, xyz_undeformed=None,
             elem_fibre=None, elem_material=None, materials=None, request_jacobian=True, pattern=None,
             e_begin=0, e_end=None, sides=None, threads=1, fast=False):
    """Reference-order whole-mesh assembly (elements [e_begin,e_end)).  Returns (row_ptr, col, val, rhs).
    threads > 1: rows split over host cores (oracle_assemble_mt, bitwise equal to the serial loop);
    fast: the -O3 -march=native timing build (cpu_baseline only)."""
    conn = np.ascontiguousarray(conn, dtype=np.uint32)
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    n_node = xyz.shape[0]
    n_owned = n_node if n_owned is None else n_owned
    e_end = conn.shape[0] if e_end is None else e_end
    if pattern is None:
        row_ptr, col, _, _ = build_pattern(elem_type, conn, n_node, n_owned, nvar)
    else:
        row_ptr, col = pattern
    val = np.zeros(col.shape[0])
    rhs = np.zeros(n_owned * nvar)
    cu = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
    u_old, aux, xyz_undeformed, elem_fibre = cu(u_old), cu(aux), cu(xyz_undeformed), cu(elem_fibre)
    em = None if elem_material is None else np.ascontiguousarray(elem_material, dtype=np.int32)
    mats = None if materials is None else (SolidMaterial * len(materials))(*materials)
    L = fast_lib() if fast else lib()
    rc = L.oracle_assemble_mt(int(threads), int(model), int(elem_type), C.c_int64(e_begin), C.c_int64(e_end), C.c_int64(n_owned),
                              _p(conn, C.c_uint32), _p(xyz), int(nvar), _p(u_old), _p(aux), _p(xyz_undeformed),
                              _p(elem_fibre), _p(em, C.c_int32), mats, C.byref(params), int(request_jacobian),
                              _p(row_ptr, C.c_int64), _p(col, C.c_int32), _p(val), _p(rhs))
    assert rc == 0, rc
    if sides is not None:
        se, si, sd = sides
        se = np.ascontiguousarray(se, dtype=np.int64)
        si = np.ascontiguousarray(si, dtype=np.int32)
        sd = np.ascontiguousarray(sd, dtype=np.float64)
        rc = lib().oracle_assemble_solid_sides(int(elem_type), C.c_int64(se.shape[0]), _p(se, C.c_int64), _p(si, C.c_int32),
                                               _p(sd), C.c_int64(n_owned), _p(conn, C.c_uint32), _p(xyz), _p(xyz_undeformed),
                                               C.byref(params), int(request_jacobian), _p(row_ptr, C.c_int64),
                                               _p(col, C.c_int32), _p(val), _p(rhs))
        assert rc == 0, rc
    return row_ptr, col, val, rhs


def solid_post_process(elem_type, conn, xyz, xyz_undeformed, elem_fibre, elem_material, materials, pseudo_time):
    """SolidSystem::post_process -> (pressure [ne], von_mises [ne], fibre_current [ne][3])"""
    conn = np.ascontiguousarray(conn, dtype=np.uint32)
    ne = conn.shape[0]
    cu = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    xyz, xu, fib = cu(xyz), cu(xyz_undeformed), cu(elem_fibre)
    em = np.ascontiguousarray(elem_material, dtype=np.int32)
    mats = (SolidMaterial * len(materials))(*materials)
    pr, vm, fc = np.empty(ne), np.empty(ne), np.empty((ne, 3))
    rc = lib().oracle_solid_post_process(int(elem_type), C.c_int64(ne), _p(conn, C.c_uint32), _p(xyz), _p(xu), _p(fib),
                                         _p(em, C.c_int32), mats, C.c_double(pseudo_time), _p(pr), _p(vm), _p(fc))
    assert rc == 0
    return pr, vm, fc


def pihna_volume_integrals(elem_type, conn, xyz, u, ranges, n_elem=None):
    conn = np.ascontiguousarray(conn, dtype=np.uint32)
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    u = np.ascontiguousarray(u, dtype=np.float64)
    out = np.zeros(4)
    ne = conn.shape[0] if n_elem is None else n_elem
    rc = lib().oracle_pihna_volume_integrals(int(elem_type), C.c_int64(ne), _p(conn, C.c_uint32), _p(xyz), _p(u), C.byref(ranges), _p(out))
    assert rc == 0
    return out


def ripf_volume_integrals(elem_type, conn, xyz, u, ranges, n_elem=None):
    conn = np.ascontiguousarray(conn, dtype=np.uint32)
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    u = np.ascontiguousarray(u, dtype=np.float64)
    out = np.zeros(2)
    ne = conn.shape[0] if n_elem is None else n_elem
    rc = lib().oracle_ripf_volume_integrals(int(elem_type), C.c_int64(ne), _p(conn, C.c_uint32), _p(xyz), _p(u), C.byref(ranges), _p(out))
    assert rc == 0
    return out


def adpm_parcellation_integrals(elem_type, conn, xyz, u, ranges, elem_subdomain, ids, n_elem=None):
    """-> [n_ids][4] = A_b concentration, Tau concentration, A_b volume, Tau volume per parcellation id"""
    conn = np.ascontiguousarray(conn, dtype=np.uint32)
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    u = np.ascontiguousarray(u, dtype=np.float64)
    sub = np.ascontiguousarray(elem_subdomain, dtype=np.int32)
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    out = np.zeros((ids.size, 4))
    ne = conn.shape[0] if n_elem is None else n_elem
    rc = lib().oracle_adpm_parcellation_integrals(int(elem_type), C.c_int64(ne), _p(conn, C.c_uint32), _p(xyz), _p(u), C.byref(ranges),
                                                  _p(sub, C.c_int32), _p(ids, C.c_int32), C.c_int32(ids.size), _p(out))
    assert rc == 0
    return out


def ripf_check_solution(params, sol, prev, rt):
    """-> (clamped solution, new prev, time derivative, rt with total, aux, RT_total_max)"""
    cp = lambda a: np.ascontiguousarray(a, dtype=np.float64).copy()
    sol, prev, rt = cp(sol), cp(prev), cp(rt)
    td, aux = np.empty_like(sol), np.empty_like(sol)
    L = lib()
    L.oracle_ripf_check_solution.restype = C.c_double
    mx = L.oracle_ripf_check_solution(C.c_int64(sol.shape[0]), C.byref(params), _p(sol), _p(prev), _p(td), _p(rt), _p(aux))
    return sol, prev, td, rt, aux, mx


def clamp_nonnegative(u):
    u = np.ascontiguousarray(u, dtype=np.float64).copy()
    lib().oracle_clamp_nonnegative(_p(u), C.c_int64(u.size))
    return u
