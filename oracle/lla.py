"""Oracle restatement of ``src/lla.py`` (TEST INFRASTRUCTURE — see ``oracle/__init__.py``)."""
from __future__ import annotations

import torch
from torch.func import jacrev, jvp

import lip_amd  # noqa: F401
from lip_amd.distributions import MultivariateNormalFullCovariance
from lip_amd.utils import flatten_nn_params

from .ggn import compute_ggn_dense, compute_ggn_vp
from .sample import sample


def compute_curvature_approx(map_state, Z, model_type, alpha, full_set_size=None):
    """``src/lla.py:11-23``: v -> GGN v + alpha v."""
    ggn_vp = compute_ggn_vp(map_state, Z, model_type=model_type, full_set_size=full_set_size)
    return lambda v: ggn_vp(v) + alpha * v


def compute_curvature_approx_dense(map_state, x, model_type, alpha, full_set_size=None):
    """``src/lla.py:26-34``."""
    GGN, flat_params_map, unravel_fn = compute_ggn_dense(map_state, x, model_type=model_type,
                                                         full_set_size=full_set_size)
    GGN = GGN + alpha * torch.eye(GGN.shape[0], dtype=GGN.dtype)
    return GGN, flat_params_map, unravel_fn


def posterior_lla_dense(map_state, x, model_type, alpha, full_set_size=None, return_unravel_fn=False):
    """``src/lla.py:37-48``."""
    S_inv, flat_params_map, unravel_fn = compute_curvature_approx_dense(
        map_state, x, model_type=model_type, alpha=alpha, full_set_size=full_set_size)
    S = torch.linalg.solve(S_inv, torch.eye(S_inv.shape[0], dtype=S_inv.dtype))
    dist = MultivariateNormalFullCovariance(loc=flat_params_map.to(torch.float64), covariance_matrix=S)
    return (dist, unravel_fn) if return_unravel_fn else dist


def _flat_apply(map_state, unravel_fn, model_type):
    def flat_apply_fn(flat_p, inputs):
        p = unravel_fn(flat_p)
        if model_type == "regressor":
            return map_state.apply_fn(p, inputs, return_logvar=False)
        variables = dict(p)
        variables["batch_stats"] = map_state.batch_stats
        return map_state.apply_fn(variables, inputs, train=False, mutable=False)
    return flat_apply_fn


def predict_lla_dense(map_state, Xnew, Z, model_type, alpha, full_set_size=None):
    """``src/lla.py:51-82``: per-test-point J S J^T (diagonal matrix for the regressor)."""
    S_inv, flat_params_map, unravel_fn = compute_curvature_approx_dense(
        map_state, Z, model_type=model_type, alpha=alpha, full_set_size=full_set_size)
    S = torch.linalg.solve(S_inv, torch.eye(S_inv.shape[0], dtype=S_inv.dtype))
    flat_apply_fn = _flat_apply(map_state, unravel_fn, model_type)
    Jnew = torch.stack([
        jacrev(lambda fp: flat_apply_fn(fp, xi[None]).squeeze())(flat_params_map) for xi in Xnew])
    f_mean = flat_apply_fn(flat_params_map, Xnew).squeeze()
    if model_type == "regressor":
        f_cov = torch.diag(torch.stack([Ji @ S @ Ji for Ji in Jnew]))       # :77
    else:
        f_cov = torch.stack([Ji @ S @ Ji.T for Ji in Jnew])
    return MultivariateNormalFullCovariance(loc=f_mean, covariance_matrix=f_cov)


def predict_lla_scalable(map_state, Xnew, Z, model_type, alpha, key=None, full_set_size=None, num_samples=1,
                         **sample_kw):
    """``src/lla.py:133-156``: f(x; theta) + J(x) w_s with w_s ~ sample(...)  -> (S, B, C)."""
    flat_params, unravel_fn = flatten_nn_params(map_state.params)
    D = flat_params.shape[0]
    key = key if key is not None else 123
    w_samples = sample(map_state, Z, D, alpha=alpha, key=key, model_type=model_type,
                       num_samples=num_samples, full_set_size=full_set_size, **sample_kw)
    model_fun = _flat_apply(map_state, unravel_fn, model_type)
    fmu = model_fun(flat_params, Xnew)
    fz = lambda p: model_fun(p, Xnew)
    dys = torch.stack([jvp(fz, (flat_params,), (w,))[1] for w in w_samples])
    return fmu[None] + dys


def materialize_covariance(f_cov_vp, N, out_dim, mode="diag"):
    """``src/lla.py:160-217``."""
    K = N * out_dim
    if mode == "diag":
        diag = torch.zeros(K, dtype=torch.float64)
        for i in range(K):
            e_i = torch.zeros(K, dtype=torch.float64)
            e_i[i] = 1.0
            diag[i] = f_cov_vp(e_i).reshape(K)[i]
        return diag.reshape(N, out_dim)
    if mode == "full":
        cov = torch.zeros(K, K, dtype=torch.float64)
        for i in range(K):
            e_i = torch.zeros(K, dtype=torch.float64)
            e_i[i] = 1.0
            cov[:, i] = f_cov_vp(e_i).reshape(K)
        return cov
    raise ValueError("mode must be 'diag' or 'full'")
