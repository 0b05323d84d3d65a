"""Restatement of the reference's ``tests/test_ggn.py`` (same names, assertions and tolerances for the
float64 oracle; fp32 tolerances stated for the HIP path)."""
import numpy as np
import torch

from fixtures import (classification_2d_data, classifier_state, regression_1d_data, small_model_state)  # noqa: F401
from impl import cpu64, impl  # noqa: F401
from lip_amd.utils import flatten_nn_params, is_pd


def test_full_ggn_vs_closed_form_hessian(impl, regression_1d_data, small_model_state):
    """reference test_full_ggn_vs_jax_hessian (:21-54): for the fixed-variance linear model the GGN equals
    the Hessian of the NLL, which is exp(-logvar) * [[sum x^2, sum x], [sum x, n]] =
    c * [[14.46, 3.6], [3.6, 4]] on X = [-1, 0, 1.1, 3.5] (the RNG-free known answer)."""
    X, y = regression_1d_data
    state = small_model_state
    GGN, *_ = impl.ggn.compute_ggn_dense(impl.state(state), impl.tensor(X), model_type="regressor")
    c = torch.exp(-state.params["logvar"]["logvar"]).item()
    full_hessian = c * np.array([[14.46, 3.6], [3.6, 4.0]])
    np.testing.assert_allclose(cpu64(GGN).numpy(), full_hessian, rtol=1e-2, atol=1e-3,
                               err_msg="GGN does not match the closed-form Hessian within tolerance")
    # and against autograd's Hessian of the total NLL, as the reference does with jax.hessian
    flat, unravel = flatten_nn_params(state.params)

    def total_nll(fp):
        p = unravel(fp)
        mu = state.apply_fn(p, X, return_logvar=False)
        lv = state.params["logvar"]["logvar"]
        return (0.5 * (torch.log(2 * torch.pi * torch.exp(lv)) + (y - mu) ** 2 / torch.exp(lv))).sum()

    H = torch.autograd.functional.hessian(total_nll, flat)
    np.testing.assert_allclose(cpu64(GGN).numpy(), H.numpy(), rtol=1e-2, atol=1e-3)


def test_full_ggn_shape(impl, regression_1d_data, small_model_state):
    X, y = regression_1d_data
    GGN, flat_params, unravel_fn = impl.ggn.compute_ggn_dense(impl.state(small_model_state), impl.tensor(X),
                                                              model_type="regressor")
    assert GGN.shape[0] == GGN.shape[1], "GGN must be square"
    assert GGN.shape[0] == flat_params.shape[0]


def test_full_ggn_pd(impl, regression_1d_data, small_model_state):
    X, y = regression_1d_data
    GGN, *_ = impl.ggn.compute_ggn_dense(impl.state(small_model_state), impl.tensor(X), model_type="regressor")
    assert is_pd(cpu64(GGN)), "GGN is not positive definite!"


def test_ggnvp_I_vs_full_ggn(impl, regression_1d_data, small_model_state):
    """reference :87-102: vmap(ggn_vp)(I) == dense GGN, atol 1e-8 (fp32 HIP: 1e-5 here, values O(10))."""
    X, y = regression_1d_data
    st, Xd = impl.state(small_model_state), impl.tensor(X)
    I = torch.eye(2, dtype=impl.dtype, device=impl.device)
    vp = impl.ggn.compute_ggn_vp(st, Xd, model_type="regressor")
    mf_GGN = impl.rows(vp, I).T                      # in_axes=1, out_axes=1
    full_GGN, *_ = impl.ggn.compute_ggn_dense(st, Xd, model_type="regressor")
    assert torch.all(torch.isclose(cpu64(mf_GGN), cpu64(full_GGN), atol=impl.tol(1e-8, 1e-5))), "GGNs don't match!"
    # parity with the oracle on the same inputs
    import oracle.ggn as og
    ref, *_ = og.compute_ggn_dense(small_model_state, X, "regressor")
    assert torch.allclose(cpu64(mf_GGN), ref, atol=impl.tol(1e-10, 1e-5))


def test_ggnvp_I_vs_full_ggn_classifier(impl, classification_2d_data, classifier_state):
    """reference :106-131 (atol 1e-6).  40 of the 200 blob points keep the literal CPU oracle fast."""
    X, y = classification_2d_data
    X = X[::5]
    import oracle.ggn as og
    full_ref, flat, _ = og.compute_ggn_dense(classifier_state, X, "classifier")
    D = flat.shape[0]
    st, Xd = impl.state(classifier_state), impl.tensor(X)
    vp = impl.ggn.compute_ggn_vp(st, Xd, model_type="classifier")
    cols = torch.arange(0, D, 1 if impl.is_hip else 29)
    I = torch.eye(D, dtype=impl.dtype, device=impl.device)[cols]
    mf = cpu64(impl.rows(vp, I))
    assert torch.all(torch.isclose(mf, full_ref[cols], atol=impl.tol(1e-6, 2e-5))), "GGNs don't match for classifier!"
    if impl.is_hip:
        full, *_ = impl.ggn.compute_ggn_dense(st, Xd, model_type="classifier")
        assert torch.allclose(cpu64(full), full_ref, atol=2e-5, rtol=1e-4)


def test_batched_cpu_baseline_equals_literal(classification_2d_data, classifier_state):
    """The example-batched CPU restatement timed by bench.py computes the same GGN-vp as the literal loop."""
    import oracle.ggn as og
    X, y = classification_2d_data
    X = X[::10]
    v = torch.randn(354, dtype=torch.float64, generator=torch.Generator().manual_seed(0))
    a = og.compute_ggn_vp(classifier_state, X, "classifier", full_set_size=77)(v)
    b = og.compute_ggn_vp_batched(classifier_state, X, "classifier", full_set_size=77)(v)
    assert torch.allclose(a, b, rtol=1e-10, atol=1e-12)
