"""Log-marginal-likelihood in the prior precision alpha — the reference's ``src/train_alpha.py`` ("next" row N2).

``log_marginal_likelihood`` (``:13-44``), ``update_alpha`` (``:47-59``), plus ``fit_alpha`` (the alpha hyper-steps of
``train_map_then_alpha`` ``:65-121`` for a fixed theta; the interleaved MAP training is out of scope).

  log p(D | alpha) = -1/2 alpha ||theta||^2 + 1/2 D log alpha - 1/2 [ log det(I_d + (N/n)/alpha W^T W) + D log alpha ]

W^T W does not depend on alpha, so after ONE Gram (d engine rows + a float64 GEMM) and ONE eigendecomposition the
value and the exact gradient w.r.t. log alpha are O(d) scalar formulas — the reference rebuilds the Gram with
d x (M VJP + M JVP) network passes inside every ``jax.grad`` call (``:32,56``).
"""
from __future__ import annotations

import dataclasses
import math
from typing import Optional, Tuple

import torch

from .ggn import build_WTW, compute_W_vps
from .utils import count_model_params, flatten_nn_params


def _spectrum(X, state, model_type):
    """eigenvalues (float64, clamped at 0) of W^T W built at N/M = 1 (``:28-32``), D, ||theta||^2"""
    W, WT = compute_W_vps(state, X, model_type, full_set_size=None)
    inner = WT.out_shape
    d = math.prod(inner)
    WTW = build_WTW(W, WT, inner, d, dtype=torch.float64, block=1)
    lam = torch.linalg.eigvalsh(WTW).clamp_min(0.0)
    flat_p, _ = flatten_nn_params(state.params)
    D = flat_p.numel()
    return lam, D, float((flat_p.double() ** 2).sum())


def _lml_from_spectrum(alpha: float, lam: torch.Tensor, D: int, theta2: float, rescale: float):
    x = rescale * lam / alpha
    logdet_lowrank = torch.log1p(x).sum()
    logdet_term = logdet_lowrank + D * math.log(alpha)                     # :36
    log_prior = -0.5 * alpha * theta2 + 0.5 * D * math.log(alpha)          # :40-42
    value = log_prior - 0.5 * logdet_term                                  # :44
    # d/d(log alpha): -1/2 alpha ||theta||^2 + 1/2 sum_i x_i / (1 + x_i)
    grad_log_alpha = -0.5 * alpha * theta2 + 0.5 * (x / (1.0 + x)).sum()
    return float(value), float(grad_log_alpha)


def log_marginal_likelihood(alpha, X, state, model_type: str, full_set_size: Optional[int] = None) -> float:
    """``src/train_alpha.py:13-44``: log p(D|alpha) up to alpha-independent constants."""
    N = full_set_size or X.shape[0]
    lam, D, theta2 = _spectrum(X, state, model_type)
    return _lml_from_spectrum(float(alpha), lam, D, theta2, N / X.shape[0])[0]


def log_marginal_likelihood_and_grad(alpha, X, state, model_type, full_set_size=None) -> Tuple[float, float]:
    """value and d/d(log alpha) (what ``jax.grad(loss_fn)(log_alpha)`` differentiates, ``:54-56``)"""
    N = full_set_size or X.shape[0]
    lam, D, theta2 = _spectrum(X, state, model_type)
    return _lml_from_spectrum(float(alpha), lam, D, theta2, N / X.shape[0])


@dataclasses.dataclass
class Adam:
    """optax.adam(lr) restated for a scalar (b1 = 0.9, b2 = 0.999, eps = 1e-8, eps_root = 0)."""
    lr: float
    b1: float = 0.9
    b2: float = 0.999
    eps: float = 1e-8

    def init(self, _param=None):
        return dict(count=0, mu=0.0, nu=0.0)

    def update(self, grad: float, opt_state):
        c = opt_state["count"] + 1
        mu = self.b1 * opt_state["mu"] + (1 - self.b1) * grad
        nu = self.b2 * opt_state["nu"] + (1 - self.b2) * grad * grad
        mu_hat, nu_hat = mu / (1 - self.b1 ** c), nu / (1 - self.b2 ** c)
        return -self.lr * mu_hat / (math.sqrt(nu_hat) + self.eps), dict(count=c, mu=mu, nu=nu)


def update_alpha(log_alpha: float, opt_state, opt: Adam, *lm_args):
    """``src/train_alpha.py:47-59``: one Adam step of gradient *ascent* on log alpha (descent on -L)."""
    _, g = log_marginal_likelihood_and_grad(math.exp(log_alpha), *lm_args)
    upd, new_state = opt.update(-g, opt_state)
    return log_alpha + upd, new_state


def fit_alpha(X, state, model_type, full_set_size=None, alpha0: float = 1.0, alpha_lr: float = 5e-2, steps: int = 200):
    """The alpha hyper-steps of ``train_map_then_alpha`` (``:76-78,91-100``) for a fixed theta; the spectrum of
    W^T W is computed once and reused by every step."""
    N = full_set_size or X.shape[0]
    lam, D, theta2 = _spectrum(X, state, model_type)
    opt = Adam(alpha_lr)
    st = opt.init()
    la = math.log(alpha0)
    history = []
    for _ in range(steps):
        v, g = _lml_from_spectrum(math.exp(la), lam, D, theta2, N / X.shape[0])
        history.append((math.exp(la), v))
        upd, st = opt.update(-g, st)
        la += upd
    return math.exp(la), history
