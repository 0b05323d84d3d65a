"""Linearised-Laplace API — same call surface as the reference's ``src/lla.py``.

``compute_curvature_approx`` (``:11``), ``compute_curvature_approx_dense`` (``:26``),
``posterior_lla_dense`` (``:37``), ``predict_lla_dense`` (``:51``), ``predict_la_samples_dense``
(``:84``), ``predict_lla_scalable`` (``:133``), ``materialize_covariance`` (``:160``).
Network derivatives come from the HIP engine; the small dense solves use torch.linalg on the device
(SURVEY K10).
"""
from __future__ import annotations

import math

import torch

from .distributions import MultivariateNormalFullCovariance
from .engine import LinearizedNet
from .ggn import BlockOperator, compute_ggn_dense, compute_ggn_vp, get_engine
from .sample import sample
from .utils import flatten_nn_params


def compute_curvature_approx(map_state, Z, model_type, alpha, full_set_size=None):
    """``src/lla.py:11-23``: v -> GGN v + alpha v (alpha fused into the engine call)."""
    vp = compute_ggn_vp(map_state, Z, model_type=model_type, full_set_size=full_set_size)
    eng = vp.engine
    M = Z.shape[0]
    N = full_set_size or M
    scale = N / M * (math.exp(-float(map_state.params["logvar"]["logvar"])) if model_type == "regressor" else 1.0)
    from .ggn import attach_quadratic_forms
    return attach_quadratic_forms(BlockOperator(lambda V: eng.ggn_vp(V, scale, float(alpha)), (eng.D,), (eng.D,), eng, "curvature_vp"),
                                  eng, scale, float(alpha))


def compute_curvature_approx_dense(map_state, x, model_type, alpha, full_set_size=None):
    """``src/lla.py:26-34``: ``(GGN + alpha I, theta_MAP, unravel_fn)``."""
    GGN, flat_params_map, unravel_fn = compute_ggn_dense(map_state, x, model_type=model_type, full_set_size=full_set_size)
    GGN = GGN + alpha * torch.eye(GGN.shape[0], device=GGN.device, dtype=GGN.dtype)
    return GGN, flat_params_map, unravel_fn


def posterior_lla_dense(map_state, x, model_type, alpha, full_set_size=None, return_unravel_fn=False):
    """``src/lla.py:37-48``: N(theta_MAP, (GGN + alpha I)^-1); the solve runs in float64 on the device
    (the reference casts the mean to float64, ``:43``)."""
    S_inv, flat_params_map, unravel_fn = compute_curvature_approx_dense(
        map_state, x, model_type=model_type, alpha=alpha, full_set_size=full_set_size)
    S_inv = S_inv.double()
    S_inv = 0.5 * (S_inv + S_inv.T)
    S = torch.linalg.solve(S_inv, torch.eye(S_inv.shape[0], device=S_inv.device, dtype=torch.float64))
    dist = MultivariateNormalFullCovariance(loc=flat_params_map.double(), covariance_matrix=S)
    return (dist, unravel_fn) if return_unravel_fn else dist


def _jacobians(map_state, Xnew, model_type):
    """Per-test-point Jacobians (B, K, D) and outputs (B, K) from an engine bound to Xnew."""
    eng = get_engine(map_state, Xnew, model_type)
    B, K = eng.n, eng.K
    U = torch.zeros(B * K, B, K, device=eng.device, dtype=torch.float32)
    idx = torch.arange(B * K, device=eng.device)
    U[idx, idx // K, idx % K] = 1.0
    J = eng.vjp(U, "raw").reshape(B, K, eng.D)
    return J, eng.outputs()


def predict_lla_dense(map_state, Xnew, Z, model_type, alpha, full_set_size=None):
    """``src/lla.py:51-82``: f_cov_i = J_i S J_i^T per test point ((B, K, K); a diagonal (B, B) matrix for
    the regressor, ``:77``)."""
    S_inv, flat_params_map, _ = compute_curvature_approx_dense(map_state, Z, model_type=model_type, alpha=alpha,
                                                               full_set_size=full_set_size)
    S_inv = S_inv.double()
    S = torch.linalg.solve(0.5 * (S_inv + S_inv.T), torch.eye(S_inv.shape[0], device=S_inv.device, dtype=torch.float64))
    J, f = _jacobians(map_state, Xnew, model_type)
    J = J.double()
    f_mean = f.double().squeeze()
    f_cov = J @ S @ J.transpose(-1, -2)                       # (B, K, K)
    if model_type == "regressor":
        f_cov = torch.diag(f_cov.reshape(-1))
    return MultivariateNormalFullCovariance(loc=f_mean, covariance_matrix=f_cov)


def predict_la_samples_dense(map_state, Xnew, Z, model_type, alpha, full_set_size=None, num_mc_samples=100, key=None):
    """``src/lla.py:84-129``: non-linearised LA — sample theta ~ N(theta_MAP, S) and run the network
    (plotting helper in the reference; the forward passes use the torch functional NetSpec forward on
    the device)."""
    S_inv, flat_params_map, unravel_fn = compute_curvature_approx_dense(map_state, Z, model_type=model_type, alpha=alpha,
                                                                        full_set_size=full_set_size)
    dev = S_inv.device
    S = torch.linalg.inv(S_inv.double())
    dist = MultivariateNormalFullCovariance(flat_params_map.double(), S)
    flat_samples = dist.sample((num_mc_samples,), seed=0 if key is None else int(key)).float()
    net = map_state.net
    stats = {k: v for k, v in map_state.batch_stats.items()} if map_state.batch_stats else {}
    from .utils import tree_map
    stats = tree_map(lambda t: torch.as_tensor(t).to(dev, torch.float32), stats)
    X = Xnew.to(dev, torch.float32)
    outs = [net.forward(unravel_fn(fp), stats, X) for fp in flat_samples]
    out = torch.stack(outs)
    return out.squeeze(-1) if model_type == "regressor" else out


def predict_lla_scalable(map_state, Xnew, Z, model_type, alpha, key=None, full_set_size=None, num_samples=1, **sample_kw):
    """``src/lla.py:133-156``: f(x; theta_MAP) + J(x) w_s, w_s = sample(...)  -> (S, B, C).
    One engine is bound to the test batch; the S JVPs run as one tangent-forward block (the reference
    maps them sequentially, ``:154``).  ``sample_kw`` goes to :func:`sample` (``reference_compat=True`` for the
    reference's clipped small-space Lanczos)."""
    flat_params, _ = flatten_nn_params(map_state.params)
    D = flat_params.shape[0]
    key = key if key is not None else 123                                       # :136
    w_samples = sample(map_state, Z, D, alpha=alpha, key=key, model_type=model_type, num_samples=num_samples,
                       full_set_size=full_set_size, **sample_kw)
    eng = get_engine(map_state, Xnew, model_type, workspace_bytes=4 << 30)      # a per-batch binding: capped workspace
    fmu = eng.outputs()                                                         # (B, C)
    dys = eng.jvp(w_samples, "raw")                                             # (S, B, C)
    return fmu[None] + dys


def _cross_gram64(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """A B^T for (a, D), (b, D) float32 factors with float64 accumulation (``lip_dot_nt_f64``)."""
    from . import krylov
    return krylov.dot_nt(A.contiguous(), B.contiguous())


def predict_lla_marginals(map_state, Xnew, Z, model_type, alpha, full_set_size=None, batch: int = 64):
    """The exact linearised predictive of ``predict_lla_dense`` (``src/lla.py:51-82``) — mean f(x; theta_MAP) and the
    K x K covariance J(x) S J(x)^T per test point, S = (alpha I + beta W W^T)^-1 — without anything D x D:
    S = alpha^-1 (I - W (alpha/beta I + W^T W)^-1 W^T)  =>  J S J^T = alpha^-1 (J J^T - (J W) C (J W)^T).
    The Jacobian rows of a test batch come from ONE per-example backward sweep of K probes (``lip_vjp_rows``), J W is a
    float64-accumulated product against the factor.  Per-point marginals are what the Monte-Carlo estimates of
    ``predict_lla_scalable`` feed into (NLL, accuracy, Brier, ECE): this gives them exactly (no sampling noise in the
    covariance).  It is not faster than the sampled route at the CIFAR config (0.86 s against 0.34 s per 256-image
    batch with 200 draws: the float64 products dominate), so it is the reference point, not the default.
    Not a reference function (its scalable predictive is sample-based only); returned like ``predict_lla_dense``."""
    from .ggn import gram_from_factor, materialize_factor
    eng_z = get_engine(map_state, Z, model_type)
    M = Z.shape[0]
    N = full_set_size or M
    beta = N / M
    c = math.exp(-0.5 * float(map_state.params["logvar"]["logvar"])) if model_type == "regressor" else 1.0
    Wm = materialize_factor(eng_z, c)                                   # (d, D)
    Gd = gram_from_factor(Wm)
    d = Wm.shape[0]
    # (alpha/beta I + Gd)^-1 restricted to range(Gd): on the null space of Gd (the classifier's factor has rank
    # M (K-1)) J W vanishes exactly, but its rounding error would be amplified by beta/alpha there
    lam, Ug = torch.linalg.eigh(0.5 * (Gd + Gd.T))
    keep = lam > 1e-6 * lam.max().clamp_min(1e-300)
    Cm = (Ug * torch.where(keep, 1.0 / (alpha / beta + lam.clamp_min(0.0)), torch.zeros_like(lam))) @ Ug.T
    means, covs = [], []
    for s0 in range(0, Xnew.shape[0], batch):
        Xb = Xnew[s0:s0 + batch]
        eng = get_engine(map_state, Xb, model_type)
        K = eng.K
        E = torch.eye(K, device=eng.device, dtype=torch.float32)[:, None, :].expand(K, eng.n, K).contiguous()
        J = eng.vjp_rows(E, "raw").permute(1, 0, 2)                     # (B, K, D)
        JJ = torch.stack([_cross_gram64(J[i], J[i]) for i in range(J.shape[0])])
        JW = _cross_gram64(J.reshape(-1, eng.D), Wm).reshape(eng.n, K, d)
        cov = (JJ - JW @ Cm @ JW.transpose(-1, -2)) / alpha
        means.append(eng.outputs().double())
        covs.append(0.5 * (cov + cov.transpose(-1, -2)))
    f_mean, f_cov = torch.cat(means), torch.cat(covs)
    if model_type == "regressor":
        return MultivariateNormalFullCovariance(loc=f_mean.squeeze(), covariance_matrix=torch.diag(f_cov.reshape(-1)))
    return MultivariateNormalFullCovariance(loc=f_mean.squeeze(), covariance_matrix=f_cov)


def materialize_covariance(f_cov_vp, N, out_dim, mode="diag"):
    """``src/lla.py:160-217``: probe an operator with the K = N*out_dim basis vectors."""
    K = N * out_dim
    I = torch.eye(K, dtype=torch.float64)
    if mode == "diag":
        return torch.stack([torch.as_tensor(f_cov_vp(I[i])).reshape(K)[i] for i in range(K)]).reshape(N, out_dim)
    if mode == "full":
        return torch.stack([torch.as_tensor(f_cov_vp(I[i])).reshape(K) for i in range(K)], dim=1)
    raise ValueError("mode must be 'diag' or 'full'")
