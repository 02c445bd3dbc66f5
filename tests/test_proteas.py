"""PROTEAS assembly (src/proteas.C:338-705), SURVEY §8(f) rank 3."""
import numpy as np
import pytest

from conftest import shim_rows
from rdcfes_amd import proteas_params_from_dict, synth


def _elem(nen, seed, variant="full"):
    rng = np.random.default_rng(seed)
    if nen == 4:
        X = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], float) + 0.15 * rng.standard_normal((4, 3))
    else:
        X = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], float)
        X = X + 0.08 * rng.standard_normal((8, 3))
    u = np.column_stack([rng.uniform(0.3, 0.7, nen), rng.uniform(0.0, 0.4, nen), rng.uniform(0.0, 0.1, nen),
                         rng.uniform(0.02, 0.2, nen), rng.uniform(0.0, 0.2, nen)])
    aux = np.column_stack([rng.uniform(5.0, 45.0, nen), rng.uniform(0.0, 60.0, nen), np.zeros(nen)])
    return X, u, aux, proteas_params_from_dict(synth.proteas_param_dict(variant))


def test_parameter_keys():
    from rdcfes_amd.params import PROTEAS_KEYS
    assert PROTEAS_KEYS["host/RT_exp_a"] == "host_RT_exp_a" and PROTEAS_KEYS["oedema/reabsorption_rate"] == "oedema_reabsorption_rate"
    assert PROTEAS_KEYS["tumour/diffusion_host"] == "tumour_diffusion_host" and len(PROTEAS_KEYS) == 28
    p = proteas_params_from_dict({})
    assert p.time_step == 1.0e-9 and p.necrosis_slope == 1.0


@pytest.mark.parametrize("nen", [4, 8])
def test_oracle_dose_quirk(oracle, nen):
    """Only AUX variable 0 at LOCAL NODE 1 reaches the integrands (RTD = phi_1 * AUX0[node 1], src/proteas.C:481)."""
    X, u, aux, p = _elem(nen, 3)
    Ke0, Fe0 = oracle.element(oracle.MODEL_PROTEAS, nen, X, u, p, aux=aux)
    a2 = aux.copy()
    a2[:, 1] = -7.0                       # "RTD" variable: never read
    a2[np.arange(nen) != 1, 0] = 123.0    # HU at the other nodes: never read
    Ke1, Fe1 = oracle.element(oracle.MODEL_PROTEAS, nen, X, u, p, aux=a2)
    np.testing.assert_array_equal(Ke0, Ke1)
    np.testing.assert_array_equal(Fe0, Fe1)
    a3 = aux.copy()
    a3[1, 0] *= 1.5
    assert np.abs(oracle.element(oracle.MODEL_PROTEAS, nen, X, u, p, aux=a3)[1] - Fe0).max() > 0


@pytest.mark.parametrize("nen", [4, 8])
def test_oracle_zero_rates(oracle, nen):
    X, u, aux, _ = _elem(nen, 5)
    zero = {k: 0.0 for k in synth.proteas_param_dict("full")}
    zero.update({"time_step": 0.05, "cells/total_capacity": 1.0, "radiotherapy/max_dosage": 1.0, "oedema/RT_exp": 1.0})
    Ke, Fe = oracle.element(oracle.MODEL_PROTEAS, nen, X, u, proteas_params_from_dict(zero), aux=aux)
    phi, dphi, jxw = oracle.fe_reinit(nen, X)
    M = np.einsum("q,qi,qj->ij", jxw, phi, phi)
    for a in range(5):
        for b in range(5):
            np.testing.assert_allclose(Ke[a * nen:(a + 1) * nen, b * nen:(b + 1) * nen], M if a == b else 0.0, atol=1e-15)
        np.testing.assert_allclose(Fe[a * nen:(a + 1) * nen], M @ u[:, a], rtol=1e-13)


@pytest.mark.parametrize("nen", [4, 8])
@pytest.mark.parametrize("variant", ["defaults", "full"])
def test_product_rows_match_oracle(oracle, shim, nen, variant):
    for seed in range(6):
        X, u, aux, p = _elem(nen, 30 + seed, variant)
        Ke0, Fe0 = oracle.element(oracle.MODEL_PROTEAS, nen, X, u, p, aux=aux)
        s = np.abs(Ke0).max()
        for fast in ((False, True) if nen == 4 else (False,)):
            Ke1, Fe1 = shim_rows(shim, 5, nen, p, X, u, aux, fast=fast)
            np.testing.assert_allclose(Ke1, Ke0, rtol=1e-10, atol=1e-13 * s)
            np.testing.assert_allclose(Fe1, Fe0, rtol=1e-10, atol=1e-13 * np.abs(Fe0).max())


@pytest.mark.gpu
@pytest.mark.parametrize("nen,n", [(4, 5), (8, 4)])
@pytest.mark.parametrize("variant", ["defaults", "full"])
@pytest.mark.parametrize("scatter", [1, 2])
@pytest.mark.parametrize("kernel_variant", [0, 1])   # 0 = auto (factored TET4 kernels), 1 = generic evaluator
def test_gpu_parity(oracle, nen, n, variant, scatter, kernel_variant):
    from rdcfes_amd import AssemblyContext, FIELD_AUX_NODAL, FIELD_OLD_SOLUTION
    conn, xyz = synth.kuhn_tet_mesh(n, jitter=0.1, order="random") if nen == 4 else synth.hex_mesh(n, jitter=0.1, order="random")
    u, aux = synth.proteas_fields(xyz)
    p = proteas_params_from_dict(synth.proteas_param_dict(variant))
    _, _, val0, rhs0 = oracle.assemble(oracle.MODEL_PROTEAS, nen, conn, xyz, 5, p, u_old=u, aux=aux)
    with AssemblyContext(0) as ctx:
        ctx.set_kernel_variant(kernel_variant)
        ctx.mesh_upload(nen, conn, xyz, 5)
        ctx.field_upload(FIELD_OLD_SOLUTION, u)
        with pytest.raises(Exception):
            ctx.assemble_proteas(p)               # aux field not set
        ctx.field_upload(FIELD_AUX_NODAL, aux)
        ctx.set_scatter(scatter)
        ctx.assemble_proteas(p)
        val, rhs = ctx.csr_download()
    assert np.linalg.norm(rhs - rhs0) <= 1e-10 * np.linalg.norm(rhs0)
    assert np.linalg.norm(val - val0) <= 1e-10 * np.linalg.norm(val0)
