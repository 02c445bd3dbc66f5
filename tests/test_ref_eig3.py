"""Pins the oracle's restatement of the eigenvalue step of SolidSystem::post_process (src/solid_system.C:509-520) against the
reference's OWN routine: src/eig3.C compiles from its own source, so it is built where it lies into oracle/_ref/libref_eig3.so
(oracle/Makefile; git-ignored, nothing of it is copied) and called here.  The hot path itself stays "parity unpinned"
(DESIGN.md §1): this covers the one step of the path's "next" rows for which the real reference can run in this container.
Also checks the invariant form the device kernel uses (k_solid_post: pressure = tr/3, von Mises = sqrt(I1^2 - 3 I2))."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def ref(oracle):
    lib = oracle.ref_lib()
    if lib is None:
        pytest.skip("oracle/_ref/libref_eig3.so not built (no /root/reference in this environment)")
    return lib


def _ref_eig(ref, A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    V, d = np.empty((3, 3)), np.empty(3)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    ref.ref_eigen_decomposition(dp(A), dp(V), dp(d))
    return d, V


def _oracle_measures(oracle, A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    ev, p, vm = np.empty(3), C.c_double(), C.c_double()
    oracle.lib().oracle_stress_measures(A.ctypes.data_as(C.POINTER(C.c_double)), ev.ctypes.data_as(C.POINTER(C.c_double)),
                                        C.byref(p), C.byref(vm))
    return ev, p.value, vm.value


def _cases():
    rng = np.random.default_rng(20241016)
    out = []
    for scale in (1.0, 1e-6, 1e6):
        for _ in range(40):
            B = rng.standard_normal((3, 3)) * scale
            out.append(0.5 * (B + B.T))
    out.append(np.zeros((3, 3)))                                   # sigma(F = I) = 0
    out.append(np.diag([2.0, 2.0, 2.0]))                           # hydrostatic: triple eigenvalue
    out.append(np.diag([5.0, -1.0, -1.0]))                         # uniaxial: double eigenvalue
    Q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    out.append(Q @ np.diag([3.0, 3.0, -7.0]) @ Q.T)                # double eigenvalue, rotated
    out.append(np.array([[1.0, 1e-9, 0.0], [1e-9, 1.0, 0.0], [0.0, 0.0, 1.0]]))   # nearly degenerate
    return out


def test_reference_eig3_reproduces_its_input(ref):
    """sanity of the door itself: A V = V diag(d), V orthonormal (columns = principal directions)"""
    for A in _cases():
        d, V = _ref_eig(ref, A)
        s = max(np.abs(A).max(), 1e-300)
        assert np.abs(A @ V - V * d[None, :]).max() <= 1e-13 * s
        assert np.abs(V.T @ V - np.eye(3)).max() <= 1e-13


def test_oracle_eigenvalues_and_stress_measures_match_the_reference(oracle, ref):
    for A in _cases():
        d, _ = _ref_eig(ref, A)
        ev, p, vm = _oracle_measures(oracle, A)
        s = max(np.abs(A).max(), 1e-300)
        assert np.abs(np.sort(ev) - np.sort(d)).max() <= 1e-13 * s
        p_ref = (d[0] + d[1] + d[2]) / 3.0                                                                    # src/solid_system.C:517
        vm_ref = np.sqrt(d[0] ** 2 + d[1] ** 2 + d[2] ** 2 - d[0] * d[1] - d[0] * d[2] - d[1] * d[2])          # :519-520
        assert abs(p - p_ref) <= 1e-13 * s
        assert abs(vm - vm_ref) <= 1e-12 * s + 1e-7 * s * (vm_ref < 1e-6 * s)   # sqrt of a cancelling sum near a triple eigenvalue


def test_invariant_form_of_the_device_kernel_matches_the_reference(ref):
    """k_solid_post takes both measures from the invariants of the averaged stress instead of an eigen-solve"""
    for A in _cases():
        d, _ = _ref_eig(ref, A)
        s = max(np.abs(A).max(), 1e-300)
        p_ref = (d[0] + d[1] + d[2]) / 3.0
        vm_ref = np.sqrt(max(d[0] ** 2 + d[1] ** 2 + d[2] ** 2 - d[0] * d[1] - d[0] * d[2] - d[1] * d[2], 0.0))
        sc = [A[0, 0], A[1, 1], A[2, 2], A[0, 1], A[1, 2], A[0, 2]]
        dev = sc[0] ** 2 + sc[1] ** 2 + sc[2] ** 2 - sc[0] * sc[1] - sc[0] * sc[2] - sc[1] * sc[2] + 3.0 * (sc[3] ** 2 + sc[4] ** 2 + sc[5] ** 2)
        assert abs((sc[0] + sc[1] + sc[2]) / 3.0 - p_ref) <= 1e-13 * s
        assert abs(np.sqrt(max(dev, 0.0)) - vm_ref) <= 1e-12 * s + 1e-7 * s * (vm_ref < 1e-6 * s)
