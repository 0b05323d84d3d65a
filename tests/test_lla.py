"""Restatement of the reference's ``tests/test_lla.py``.  The reference file is stale against its own source
(it passes ``return_Hinv=`` and unpacks ``predict_lla_scalable`` as ``(mean, cov_vp)``); the
assertions are kept, the calls follow the current signatures (``src/lla.py:26,133``)."""
import numpy as np
import torch

from fixtures import (classification_2d_data, classifier_state, regression_1d_data, sine_data,  # noqa: F401
                      small_model_state, toyregressor_state)
from impl import cpu64, impl  # noqa: F401
import oracle.lla as olla


def test_posterior_lla(impl, small_model_state, regression_1d_data):
    """reference :12-25"""
    X, y = regression_1d_data
    st, Xd = impl.state(small_model_state), impl.tensor(X)
    post = impl.lla.posterior_lla_dense(st, Xd, alpha=1.0, model_type="regressor")
    _, flat_map, _ = impl.lla.compute_curvature_approx_dense(st, Xd, alpha=1.0, model_type="regressor")
    np.testing.assert_allclose(cpu64(post.mean()).numpy(), cpu64(flat_map).numpy(), rtol=1e-4, atol=1e-6)
    assert torch.all(torch.linalg.eigvals(cpu64(post.covariance())).real > 0)
    ref = olla.posterior_lla_dense(small_model_state, X, alpha=1.0, model_type="regressor")
    assert torch.allclose(cpu64(post.covariance()), ref.covariance(), rtol=impl.tol(1e-10, 1e-4), atol=impl.tol(1e-12, 1e-6))


def test_predict_lla(impl, small_model_state, regression_1d_data):
    """reference :28-49"""
    X, y = regression_1d_data
    st, Xd = impl.state(small_model_state), impl.tensor(X)
    xnew = impl.tensor(torch.tensor([[-0.5], [0.5]], dtype=torch.float64))
    pred = impl.lla.predict_lla_dense(st, xnew, Xd, model_type="regressor", alpha=1.0)
    W, b = small_model_state.params["params"]["W"], small_model_state.params["params"]["b"]
    pred_mean = (cpu64(xnew) * W + b).squeeze(-1)
    np.testing.assert_allclose(cpu64(pred.mean()).numpy(), pred_mean.numpy(), rtol=1e-4, atol=1e-6)
    assert torch.all(torch.linalg.eigvals(cpu64(pred.covariance())).real > 0)
    ref = olla.predict_lla_dense(small_model_state, cpu64(xnew), X, "regressor", 1.0)
    assert torch.allclose(cpu64(pred.covariance()), ref.covariance(), rtol=impl.tol(1e-10, 1e-4), atol=impl.tol(1e-12, 1e-6))


def test_predict_lla_jvp(impl, toyregressor_state, sine_data):
    """reference :52-80: dense vs matrix-free predictive covariance.  ``predict_lla_scalable`` returns samples
    (S, B, C) (src/lla.py:156), so the matrix-free covariance is the sample covariance; S = 4000 on the GPU."""
    X, y = sine_data
    st, Xd = impl.state(toyregressor_state), impl.tensor(X)
    xnew = impl.tensor(torch.tensor([[-.5], [.5]], dtype=torch.float64))
    pred = impl.lla.predict_lla_dense(st, xnew, Xd, model_type="regressor", alpha=1.0)
    ref = olla.predict_lla_dense(toyregressor_state, cpu64(xnew), X, "regressor", 1.0)
    assert torch.allclose(cpu64(pred.covariance()), ref.covariance(), rtol=impl.tol(1e-9, 2e-3), atol=impl.tol(1e-12, 1e-6))
    assert torch.allclose(cpu64(pred.mean()), ref.mean(), rtol=impl.tol(1e-9, 1e-4), atol=1e-6)
    S = 4000 if impl.is_hip else 60
    kw = {} if impl.is_hip else dict(gram_rtol=1e-6, clip_min=None)
    fs = cpu64(impl.lla.predict_lla_scalable(st, xnew, Xd, model_type="regressor", alpha=1.0, num_samples=S, **kw))
    assert fs.shape == (S, 2, 1)
    var_mf = fs.squeeze(-1).var(0)
    var_dense = torch.diagonal(ref.covariance())
    rtol = 0.15 if impl.is_hip else 0.8
    assert torch.allclose(var_mf, var_dense, rtol=rtol), (var_mf, var_dense)
    assert torch.allclose(fs.squeeze(-1).mean(0), ref.mean(), atol=4 * var_dense.sqrt().max().item() / S ** 0.5 + 1e-3)


def test_predict_lla_jvp_classifier(impl, classifier_state, classification_2d_data):
    """reference :83-111: the classifier predictive covariance is symmetric PSD."""
    X, y = classification_2d_data
    X = X[::10]
    st, Xd = impl.state(classifier_state), impl.tensor(X)
    xnew = impl.tensor(torch.tensor([[-.5, .5], [1.0, -1.5], [2.0, 2.0]], dtype=torch.float64))
    pred = impl.lla.predict_lla_dense(st, xnew, Xd, alpha=0.5, model_type="classifier")
    cov = cpu64(pred.covariance())                      # (3, 2, 2)
    assert torch.allclose(cov, cov.transpose(-1, -2), rtol=1e-4, atol=1e-7)
    assert torch.all(torch.linalg.eigvalsh(0.5 * (cov + cov.transpose(-1, -2))) > -1e-6)
    ref = olla.predict_lla_dense(classifier_state, cpu64(xnew), X, "classifier", 0.5)
    assert torch.allclose(cov, ref.covariance(), rtol=impl.tol(1e-9, 2e-3), atol=impl.tol(1e-12, 1e-5))
    if impl.is_hip:
        fs = cpu64(impl.lla.predict_lla_scalable(st, xnew, Xd, model_type="classifier", alpha=0.5, num_samples=3000,
                                                 method="eigh"))
        emp = torch.stack([torch.cov(fs[:, b, :].T) for b in range(3)])
        assert torch.allclose(emp, ref.covariance(), rtol=0.25, atol=0.05 * ref.covariance().abs().max().item())
