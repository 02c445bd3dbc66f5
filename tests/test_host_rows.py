"""The product's row evaluators (rdc_row.h generic, rdc_tet4_fast.h factored), compiled for the host
by tests/host_shim.cpp, against the oracle's literal quadrature loops.  No GPU needed; the HIP
kernels run the very same functions."""
import ctypes as C

import numpy as np
import pytest

from conftest import shim_rows
from rdcfes_amd import hcc_params_from_dict, pihna_params_from_dict, ripf_params_from_dict, synth

TET = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], float)
HEX = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], float)


def _case(model, nen, seed, variant):
    rng = np.random.default_rng(seed)
    X = (TET if nen == 4 else HEX) * rng.uniform(0.5, 2.0, 3) + 0.08 * rng.standard_normal((nen, 3))
    aux = None
    if model == 0:
        p = pihna_params_from_dict(synth.pihna_param_dict(variant))
        u = np.column_stack([rng.uniform(0, 500, nen), rng.uniform(0, 2e3, nen), rng.uniform(0, 2e3, nen),
                             rng.uniform(3e3, 7e3, nen), rng.uniform(0, 1e-8, nen)])
    elif model == 1:
        p = ripf_params_from_dict(synth.ripf_param_dict(variant))
        u = np.column_stack([rng.uniform(-1000, 1000, nen), rng.uniform(0, 1, nen) * (rng.random(nen) < 0.7),
                             rng.uniform(0, 0.3, nen)])
        aux = np.column_stack([rng.uniform(-1e-2, 1e-2, nen), rng.uniform(-1e-2, 1e-2, nen), rng.uniform(0, 70, nen)])
    else:
        p = hcc_params_from_dict(synth.hcc_param_dict(variant))
        u = rng.uniform(0, 0.3, (nen, 3))
    return X, u, aux, p


CASES = [(0, "shipped"), (0, "full"), (0, "realexp"), (1, "shipped"), (1, "full"), (2, "full"), (2, "shipped")]


@pytest.mark.parametrize("model,variant", CASES)
@pytest.mark.parametrize("nen", [4, 8])
def test_generic_row_matches_oracle(oracle, shim, model, variant, nen):
    for seed in range(3):
        X, u, aux, p = _case(model, nen, seed, variant)
        Ke0, Fe0 = oracle.element(model, nen, X, u, p, aux=aux)
        Ke1, Fe1 = shim_rows(shim, model, nen, p, X, u, aux)
        s = np.abs(Ke0).max()
        np.testing.assert_allclose(Ke1, Ke0, rtol=1e-11, atol=1e-13 * s)
        np.testing.assert_allclose(Fe1, Fe0, rtol=1e-11, atol=1e-13 * np.abs(Fe0).max())


@pytest.mark.parametrize("model,variant", CASES)
def test_tet4_factored_row_matches_oracle(oracle, shim, model, variant):
    for seed in range(5):
        X, u, aux, p = _case(model, 4, 100 + seed, variant)
        Ke0, Fe0 = oracle.element(model, 4, X, u, p, aux=aux)
        Ke1, Fe1 = shim_rows(shim, model, 4, p, X, u, aux, fast=True)
        s = np.abs(Ke0).max()
        np.testing.assert_allclose(Ke1, Ke0, rtol=1e-10, atol=1e-12 * s)
        np.testing.assert_allclose(Fe1, Fe0, rtol=1e-10, atol=1e-12 * np.abs(Fe0).max())


def test_integer_exponent_shortcut_equals_pow(shim):
    """EXP_MODE 3 (x*x*x) against the general pow() instantiation of the same code."""
    X, u, aux, p = _case(0, 4, 7, "full")
    a = shim_rows(shim, 0, 4, p, X, u, fast=True)
    b = shim_rows(shim, 0, 4, p, X, u, fast=True, force_general_pow=True)
    np.testing.assert_allclose(a[0], b[0], rtol=1e-13, atol=1e-15 * np.abs(b[0]).max())
    np.testing.assert_allclose(a[1], b[1], rtol=1e-13)


def test_half_integer_exponent_shortcut_equals_pow(shim):
    """EXP_MODE 25 (x^2.5 = x*x*sqrt(x), the shipped RIPF exponent) against the general pow() instantiation."""
    for nen in (4, 8):
        X, u, aux, p = _case(1, nen, 9, "shipped")
        assert p.VolFr_exponent == 2.5
        for fast in ((True, False) if nen == 4 else (False,)):
            a = shim_rows(shim, 1, nen, p, X, u, aux, fast=fast)
            b = shim_rows(shim, 1, nen, p, X, u, aux, fast=fast, force_general_pow=True)
            np.testing.assert_allclose(a[0], b[0], rtol=1e-12, atol=1e-14 * np.abs(b[0]).max())
            np.testing.assert_allclose(a[1], b[1], rtol=1e-12)


@pytest.mark.parametrize("model,variant", CASES)
def test_structural_masks_cover_all_nonzeros(shim, model, variant):
    rng = np.random.default_rng(11)
    _, u, aux, p = _case(model, 4, 3, variant)
    w = C.c_double(-1.0)
    for n in range(4):
        a = None if aux is None else np.ascontiguousarray(aux[n])
        un = np.ascontiguousarray(u[n])
        rc = shim.shim_masks(model, C.byref(p), un.ctypes.data_as(C.POINTER(C.c_double)),
                             None if a is None else a.ctypes.data_as(C.POINTER(C.c_double)), C.byref(w))
        assert rc == 0 and w.value == 0.0


def test_pihna_nan_path_matches_reference_semantics(oracle, shim):
    """c+h+v == 0 at a quadrature point: Ve_ = 0/0 = NaN falls through both comparisons
    (src/pihna.C:477-498, App. D.5); both implementations must propagate the same NaN pattern."""
    p = pihna_params_from_dict(synth.pihna_param_dict("shipped"))
    u = np.zeros((4, 5))
    u[:, 0] = 1.0
    Ke0, Fe0 = oracle.element(0, 4, TET, u, p)
    for fast in (False, True):
        Ke1, Fe1 = shim_rows(shim, 0, 4, p, TET, u, fast=fast)
        np.testing.assert_array_equal(np.isnan(Ke1), np.isnan(Ke0))
        np.testing.assert_array_equal(np.isnan(Fe1), np.isnan(Fe0))
        assert np.isnan(Fe0).any()


def test_threshold_branches(oracle, shim):
    """crowding saturation (Te >= 1 -> Tau = 0), Te <= 0 and the cells_min_capacity switches."""
    d = synth.pihna_param_dict("full")
    p = pihna_params_from_dict(d)
    rng = np.random.default_rng(12)
    X = TET + 0.05 * rng.standard_normal((4, 3))
    for u in (np.tile([1e5, 1e5, 1e5, 1e5, 1e-9], (4, 1)),        # Te > 1
              np.tile([0.0, 0.5, 0.5, 0.9, 0.0], (4, 1)),          # below cells_min_capacity
              np.column_stack([rng.uniform(0, 2, 4)] * 4 + [rng.uniform(0, 1e-8, 4)])):  # straddles Lambda_k = 1
        Ke0, Fe0 = oracle.element(0, 4, X, u, p)
        for fast in (False, True):
            Ke1, Fe1 = shim_rows(shim, 0, 4, p, X, u, fast=fast)
            np.testing.assert_allclose(Ke1, Ke0, rtol=1e-10, atol=1e-12 * np.abs(Ke0).max())
            np.testing.assert_allclose(Fe1, Fe0, rtol=1e-10, atol=1e-12 * max(np.abs(Fe0).max(), 1e-300))


def test_pihna_cell_transport_off_variant(oracle, shim):
    """PihnaNoCellTransport (smaller structural masks) is exact when diffuse/c, taxis/c, diffuse/h, taxis/h, taxis/v
    are zero -- the shipped input -- and is refused otherwise."""
    import ctypes as C
    for seed in range(4):
        X, u, aux, p = _case(0, 4, 200 + seed, "shipped")
        Ke0, Fe0 = oracle.element(0, 4, X, u, p)
        Ke1, Fe1 = shim_rows(shim, 3, 4, p, X, u, fast=True)
        np.testing.assert_allclose(Ke1, Ke0, rtol=1e-10, atol=1e-12 * np.abs(Ke0).max())
        np.testing.assert_allclose(Fe1, Fe0, rtol=1e-10, atol=1e-12 * np.abs(Fe0).max())
    # masks cover every non-zero coefficient for such parameters
    w = C.c_double(-1.0)
    un = np.ascontiguousarray(u[0])
    assert shim.shim_masks(3, C.byref(p), un.ctypes.data_as(C.POINTER(C.c_double)), None, C.byref(w)) == 0 and w.value == 0.0
    # with any cell transport term on, the variant must not be used
    X, u, aux, pf = _case(0, 4, 1, "full")
    acc, fe = np.empty((5, 5, 4)), np.empty(5)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    Xc, uc = np.ascontiguousarray(X), np.ascontiguousarray(u)
    assert shim.shim_row(3, 4, 1, 0, C.byref(pf), dp(Xc), dp(uc), None, 0, dp(acc), dp(fe)) == 3


@pytest.mark.parametrize("general_pow", [False, True])
def test_pihna_moment_form_equals_coefficient_form(oracle, shim, general_pow):
    """PihnaNoCellTransportMoments (rdc_tet4_pihna_moments.h): the rows assembled from weighted moments of the point
    functions are the sums of the coefficient form in another association -- against the oracle at the parity
    tolerance and against the coefficient form much tighter, over states that exercise every branch (crowding
    saturated / empty, vascular fraction clamped at 0 and 1, vasculature below the diffusion threshold)."""
    import ctypes as C
    rng = np.random.default_rng(77)
    for seed in range(12):
        X, u, aux, p = _case(0, 4, 300 + seed, "shipped")
        if seed % 4 == 1:
            u[:, 3] = rng.uniform(0.0, 2.0 * p.cells_min_capacity, 4)        # v around the diffusion threshold
        if seed % 4 == 2:
            u[:, :4] *= 40.0                                                  # crowding saturated (Te >= 1) at some points
        if seed % 4 == 3:
            u[rng.integers(0, 4), 3] = 0.0                                    # Ve clamps
            u[:, 1:3] *= 1e-3
        if seed == 8:
            u[:] = 0.0                                                        # empty element: 0/0 in Ve, as upstream
        Ke0, Fe0 = oracle.element(0, 4, X, u, p)
        Ke3, Fe3 = shim_rows(shim, 3, 4, p, X, u, fast=True, force_general_pow=general_pow)
        Ke6, Fe6 = shim_rows(shim, 6, 4, p, X, u, fast=True, force_general_pow=general_pow)
        sK, sF = np.nanmax(np.abs(Ke0)), max(np.nanmax(np.abs(Fe0)), 1e-300)   # NaN (0/0 in Ve) must appear in the same places
        np.testing.assert_allclose(Ke6, Ke0, rtol=1e-10, atol=1e-12 * sK)
        np.testing.assert_allclose(Fe6, Fe0, rtol=1e-10, atol=1e-12 * sF)
        np.testing.assert_allclose(Ke6, Ke3, rtol=1e-12, atol=1e-14 * sK)
        np.testing.assert_allclose(Fe6, Fe3, rtol=1e-12, atol=1e-14 * sF)
    X, u, aux, pf = _case(0, 4, 1, "full")      # any cell transport term on: refused, like the coefficient-form variant
    acc, fe = np.empty((5, 5, 4)), np.empty(5)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    assert shim.shim_row(6, 4, 1, 0, C.byref(pf), dp(np.ascontiguousarray(X)), dp(np.ascontiguousarray(u)), None, 0, dp(acc), dp(fe), None) == 3


@pytest.mark.parametrize("general_pow", [False, True])
def test_ripf_reduced_variant(oracle, shim, general_pow):
    """RipfReduced (kappa = omicro = radiotaxis = 0 and zero HU/phi rates: the shipped run/RIPF133/input.dat) against the
    oracle's full upstream formulas and against the general Ripf rows, over states in every branch of the
    fibrosis source (HU above / inside / below the ramp, fb outside [0,1), saturated volume fraction), and refused
    when any of the dropped terms is switched on."""
    import ctypes as C
    rng = np.random.default_rng(3)
    for seed in range(12):
        X, u, aux, p = _case(1, 4, 500 + seed, "shipped")
        if seed % 4 == 1:
            u[:, 0] = rng.uniform(-1500.0, -900.0, 4)       # HU below the ramp
        if seed % 4 == 2:
            u[:, 0] = rng.uniform(10.0, 300.0, 4)           # HU >= 0: no source
            u[:, 2] = rng.uniform(0.9, 1.2, 4)              # fb around 1
        if seed % 4 == 3:
            u[:, 1] = rng.uniform(0.3, 0.6, 4)              # volume fraction saturates at some points
        Ke0, Fe0 = oracle.element(1, 4, X, u, p, aux=aux)
        Ke1, Fe1 = shim_rows(shim, 1, 4, p, X, u, A=aux, fast=True, force_general_pow=general_pow)
        Ke7, Fe7 = shim_rows(shim, 7, 4, p, X, u, A=aux, fast=True, force_general_pow=general_pow)
        sK, sF = np.abs(Ke0).max(), max(np.abs(Fe0).max(), 1e-300)
        np.testing.assert_allclose(Ke7, Ke0, rtol=1e-10, atol=1e-12 * sK)
        np.testing.assert_allclose(Fe7, Fe0, rtol=1e-10, atol=1e-12 * sF)
        np.testing.assert_allclose(Ke7, Ke1, rtol=1e-13, atol=1e-15 * sK)
        np.testing.assert_allclose(Fe7, Fe1, rtol=1e-13, atol=1e-15 * sF)
    w = C.c_double(-1.0)
    un, an = np.ascontiguousarray(u[0]), np.ascontiguousarray(aux[0])
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    assert shim.shim_masks(7, C.byref(p), dp(un), dp(an), C.byref(w)) == 0 and w.value == 0.0
    X, u, aux, pf = _case(1, 4, 1, "full")
    acc, fe = np.empty((3, 3, 4)), np.empty(3)
    assert shim.shim_row(7, 4, 1, 0, C.byref(pf), dp(np.ascontiguousarray(X)), dp(np.ascontiguousarray(u)),
                         dp(np.ascontiguousarray(aux)), 0, dp(acc), dp(fe), None) == 3


@pytest.mark.parametrize("nen", [4, 8])
def test_shipped_pattern_variants_of_hcc_and_adpm(oracle, shim, nen):
    """HccMassOnly (run/Coupled/HCC: every rate zero) and AdpmDecayOnly (run/HCP102513: decay terms only) against the
    oracle's full formulas and against the general rows; refused for parameters with any dropped term on."""
    import ctypes as C
    from rdcfes_amd import adpm_params_from_dict
    dp = lambda a: None if a is None else np.ascontiguousarray(a).ctypes.data_as(C.POINTER(C.c_double))
    for seed in range(6):
        X, u, aux, p = _case(2, nen, 700 + seed, "shipped")
        Ke0, Fe0 = oracle.element(2, nen, X, u, p)
        for fast in ((False, True) if nen == 4 else (False,)):
            Ke8, Fe8 = shim_rows(shim, 8, nen, p, X, u, fast=fast)
            Ke2, Fe2 = shim_rows(shim, 2, nen, p, X, u, fast=fast)
            np.testing.assert_allclose(Ke8, Ke0, rtol=1e-10, atol=1e-13 * np.abs(Ke0).max())
            np.testing.assert_allclose(Fe8, Fe0, rtol=1e-10, atol=1e-13 * np.abs(Fe0).max())
            np.testing.assert_allclose(Ke8, Ke2, rtol=1e-13, atol=1e-15 * np.abs(Ke0).max())
        rng = np.random.default_rng(800 + seed)
        pa = adpm_params_from_dict(synth.adpm_param_dict("shipped"), time=2.0)
        ua = np.column_stack([rng.uniform(0.0, 12.0, nen), rng.uniform(0.0, 0.02, nen), rng.uniform(0.0, 0.001, nen)])
        t = 0.1 * rng.standard_normal(3)
        Ka0, Fa0 = oracle.element(oracle.MODEL_ADPM, nen, X, ua, pa, elem_data=t)
        for fast in ((False, True) if nen == 4 else (False,)):
            Ka9, Fa9 = shim_rows(shim, 9, nen, pa, X, ua, elem_data=t, fast=fast)
            Ka4, Fa4 = shim_rows(shim, 4, nen, pa, X, ua, elem_data=t, fast=fast)
            np.testing.assert_allclose(Ka9, Ka0, rtol=1e-10, atol=1e-13 * np.abs(Ka0).max())
            np.testing.assert_allclose(Fa9, Fa0, rtol=1e-10, atol=1e-13 * np.abs(Fa0).max())
            np.testing.assert_allclose(Ka9, Ka4, rtol=1e-13, atol=1e-15 * np.abs(Ka0).max())
    X, u, aux, pf = _case(2, nen, 1, "full")
    acc, fe = np.empty((3, 3, nen)), np.empty(3)
    assert shim.shim_row(8, nen, 0, 0, C.byref(pf), dp(X), dp(u), None, 0, dp(acc), dp(fe), None) == 3
    paf = adpm_params_from_dict(synth.adpm_param_dict("full"), time=2.0)
    assert shim.shim_row(9, nen, 0, 0, C.byref(paf), dp(X), dp(u), None, 0, dp(acc), dp(fe), dp(np.zeros(3))) == 3
