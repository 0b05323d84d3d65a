"""Inducing-point KL objective — VALUES of the reference's ``src/train_inducing.py`` objectives ("next" row N1).

``alternative_objective_dense`` (``:176-193``), ``alternative_objective_scalable_exact`` (``:26-84``) and
``alternative_objective_scalable`` (``:87-173``): KL[q(theta|Z) || q(theta|data)] up to constants =
log-det term + trace term.  What is NOT here yet is the gradient w.r.t. Z (``value_and_grad`` at ``:195-196``,
``optimize_step`` ``:199-232``): it needs second-order differentiation through the network and reverse mode
through the Krylov solvers; ``variational_grad_*`` raise ``NotImplementedError``.

SURVEY §4.1-9: the reference's stochastic log-det omits beta (it bidiagonalises v -> [sqrt(alpha) v ; Wz^T v],
``:164-169``, i.e. log|alpha I + Wz Wz^T|) while its exact twin uses beta (``:68``).  ``logdet_beta=True`` (default)
is the mathematics, ``False`` the reference's behaviour.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import krylov
from .ggn import BlockOperator, build_WTW, build_WTWz, compute_W_vps
from .lla import compute_curvature_approx, compute_curvature_approx_dense
from .stochtrace import hutchpp_v2
from .utils import flatten_nn_params


def _D(state):
    return flatten_nn_params(state.params)[0].numel()


def alternative_objective_dense(Z, X, state, alpha, model_type, key=None, full_set_size=None):
    """``src/train_inducing.py:176-193`` (toy sizes: D x D matrices)."""
    S, *_ = compute_curvature_approx_dense(state, X, alpha=alpha, model_type=model_type, full_set_size=full_set_size)
    S_z, *_ = compute_curvature_approx_dense(state, Z, alpha=alpha, model_type=model_type, full_set_size=full_set_size)
    S, S_z = S.double(), S_z.double()
    S_z_inv = torch.linalg.inv(0.5 * (S_z + S_z.T))
    trace_term = torch.trace(S @ S_z_inv)
    _, S_z_inv_logdet = torch.linalg.slogdet(S_z_inv)
    return float(-S_z_inv_logdet + trace_term)


def alternative_objective_scalable_exact(Z, X, state, alpha, model_type, key=None, full_set_size=None, **_):
    """``src/train_inducing.py:26-84``: exact value through the small (d_z x d_z) and (d x d_z) Gram matrices."""
    N = full_set_size
    M, Kb = Z.shape[0], X.shape[0]
    beta, gamma = N / M, N / Kb
    alpha_inv, beta_inv = 1.0 / alpha, 1.0 / beta
    D = _D(state)
    Wz, WzT = compute_W_vps(state, Z, model_type=model_type, full_set_size=None)
    W, WT = compute_W_vps(state, X, model_type=model_type, full_set_size=None)
    inner = WzT.out_shape
    d_z = math.prod(inner)
    WzTWz = build_WTW(Wz, WzT, inner, d_z, dtype=torch.float64, block=1)
    I = torch.eye(d_z, dtype=torch.float64, device=WzTWz.device)
    _, logdet_WTW = torch.linalg.slogdet(I + beta * alpha_inv * WzTWz)
    logdet_term = logdet_WTW + D * math.log(alpha)
    d = math.prod(WT.out_shape)
    WTWz = build_WTWz(WT, Wz, inner, d=d, dtype=torch.float64, block=1)
    Mm = beta_inv * I + alpha_inv * WzTWz
    L = torch.linalg.cholesky(0.5 * (Mm + Mm.T))
    S1 = torch.cholesky_solve(WzTWz, L)
    S2 = torch.cholesky_solve(WTWz.T, L)
    trace1 = torch.trace(S1)
    trace2 = (WTWz * S2.T).sum()
    trace_term = -alpha_inv * trace1 - gamma * alpha_inv ** 2 * trace2
    return float(logdet_term + trace_term)


def alternative_objective_scalable(Z, X, state, alpha, model_type, key, full_set_size=None, st_samples=256,
                                   slq_samples=2, slq_num_matvecs: Optional[int] = None, logdet_beta: bool = True,
                                   probes: Optional[torch.Tensor] = None, return_terms: bool = False):
    """``src/train_inducing.py:87-173``: trace of S S_z^-1 by Hutch++ (s1 = st_samples - 16, s2 = 16; S_z^-1 through
    Woodbury with a dense d_z solve) + log det S_z by stochastic Lanczos quadrature on the Golub-Kahan
    bidiagonalisation of v -> [sqrt(alpha) v ; sqrt(beta) Wz^T v].  All D-space work runs on the HIP engine /
    Krylov kernels, probe blocks at a time."""
    N = full_set_size
    M = Z.shape[0]
    beta = N / M
    alpha_inv, beta_inv = 1.0 / alpha, 1.0 / beta
    S_vp = compute_curvature_approx(state, X, alpha=alpha, model_type=model_type, full_set_size=N)
    Wz, WzT = compute_W_vps(state, Z, model_type=model_type, full_set_size=None)
    eng = Wz.engine
    D, dev = eng.D, eng.device
    inner = WzT.out_shape
    d_z = math.prod(inner)
    WzTWz = build_WTW(Wz, WzT, inner, d_z, dtype=torch.float64, block=1)
    I = torch.eye(d_z, dtype=torch.float64, device=dev)
    Minv = torch.linalg.inv(beta_inv * I + alpha_inv * WzTWz).float()

    def Sz_inv_rows(V):                                     # Woodbury, :127-132
        u = WzT.rows(V).reshape(V.shape[0], d_z)
        x = (u @ Minv).reshape((V.shape[0],) + inner)
        return krylov.axpby(Wz.rows(x), V.contiguous(), None, alpha_inv, None, -alpha_inv ** 2)

    composite = BlockOperator(lambda V: S_vp.rows(Sz_inv_rows(V)), (D,), (D,), eng, "S Sz^-1")
    if probes is None:
        probes = krylov.fill_rademacher(st_samples, D, int(key), dev)          # same probes for both terms, :139-142
    st_samples = probes.shape[0]
    trace_term = float(hutchpp_v2(composite, lambda _: probes, s1=st_samples - 16, s2=16))

    k = slq_num_matvecs if slq_num_matvecs is not None else int(M * 0.8)       # :148
    sa, sb = math.sqrt(alpha), (math.sqrt(beta) if logdet_beta else 1.0)

    def A(V):                                               # :164-167  v -> [sqrt(alpha) v ; Wz^T v]
        return torch.cat([sa * V, sb * WzT.rows(V).reshape(V.shape[0], d_z)], dim=1).contiguous()

    def AT(U):
        out = Wz.rows((sb * U[:, D:]).reshape((U.shape[0],) + inner).contiguous())
        return krylov.axpby(out, U[:, :D].contiguous(), None, sa, None, 1.0)

    q = krylov.slq_logdet_product(A, AT, probes[:slq_samples].contiguous(), k, D + d_z)
    logdet_term = float(q.mean())
    if return_terms:
        return logdet_term + trace_term, logdet_term, trace_term
    return logdet_term + trace_term


def _no_grad(*_a, **_k):
    raise NotImplementedError(
        "the gradient of the KL objective w.r.t. the inducing points Z (src/train_inducing.py:195-232) is not "
        "implemented yet: it needs second-order differentiation through the network; only objective values are")


variational_grad_dense = _no_grad
variational_grad_scalable = _no_grad
optimize_step = _no_grad
train_inducing_points = _no_grad
