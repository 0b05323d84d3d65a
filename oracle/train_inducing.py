"""Oracle restatement of the objective VALUES of ``src/train_inducing.py`` (TEST INFRASTRUCTURE).
``alternative_objective_scalable`` (``:87-173``) literal: Woodbury S_z^-1, ``hutchpp_v2`` on S S_z^-1, SLQ log-det by
``integrand_funm_product_logdet(bidiag(k))`` averaged over the first ``slq_samples`` probes.  The bidiagonalisation
log-det is "parity unpinned" against the reference (no reference fixture exercises it); here it is checked against
the exact slogdet."""
import math

import torch

import lip_amd  # noqa: F401
from lip_amd.utils import flatten_nn_params

from .ggn import build_WTW, compute_W_vps
from .lla import compute_curvature_approx
from .matfree import bidiag, integrand_funm_product_logdet
from .stochtrace import hutchpp_v2


def alternative_objective_scalable(Z, X, state, alpha, model_type, probes, full_set_size, slq_samples=2,
                                   slq_num_matvecs=None, logdet_beta=True):
    ld, tr = objective_stochastic_t(Z, X, state, alpha, model_type, probes, full_set_size, slq_samples, slq_num_matvecs,
                                    logdet_beta)
    return float(ld + tr), float(ld), float(tr)


def objective_stochastic_t(Z, X, state, alpha, model_type, probes, full_set_size, slq_samples=2, slq_num_matvecs=None,
                           logdet_beta=True):
    """``alternative_objective_scalable`` (``:87-173``) as differentiable torch scalars ``(logdet_term, trace_term)``:
    ``torch.autograd.grad`` of their sum w.r.t. Z is what ``jax.value_and_grad`` (``:196``) returns for the same
    probes — reverse mode through Hutch++ (QR included) and through the re-orthogonalised bidiagonalisation."""
    N = full_set_size
    M = Z.shape[0]
    beta = N / M
    alpha_inv, beta_inv = 1.0 / alpha, 1.0 / beta
    flat, _ = flatten_nn_params(state.params)
    D = flat.numel()
    S_vp = compute_curvature_approx(state, X, alpha=alpha, model_type=model_type, full_set_size=N)
    Wz, WzT = compute_W_vps(state, Z, model_type=model_type, full_set_size=None)
    dummy = WzT(torch.zeros(D, dtype=flat.dtype))
    inner, d_z = tuple(dummy.shape), dummy.numel()
    WzTWz = build_WTW(Wz, WzT, inner, d_z, dtype=flat.dtype, block=1)
    I = torch.eye(d_z, dtype=flat.dtype)

    def Sz_inv(v):
        u = WzT(v).reshape(d_z)
        x = torch.linalg.solve(beta_inv * I + alpha_inv * WzTWz, u)
        return alpha_inv * v - alpha_inv ** 2 * Wz(x.reshape(inner))

    st = probes.shape[0]
    trace_term = hutchpp_v2(lambda v: S_vp(Sz_inv(v)), lambda _: probes, s1=st - 16, s2=16)
    k = slq_num_matvecs if slq_num_matvecs is not None else int(M * 0.8)
    sa, sb = math.sqrt(alpha), (math.sqrt(beta) if logdet_beta else 1.0)
    A = lambda v: torch.cat([sa * v, sb * WzT(v).reshape(d_z)])
    AT = lambda u: sa * u[:D] + Wz((sb * u[D:]).reshape(inner))
    quad = integrand_funm_product_logdet(bidiag(k))
    logdet_term = torch.stack([quad(A, AT, p) for p in probes[:slq_samples]]).mean()
    return logdet_term, trace_term


def variational_grad_stochastic(Z, X, state, alpha, model_type, probes, full_set_size, **kw):
    """(value, d value / dZ) of the stochastic objective on fixed probes (``src/train_inducing.py:196``)."""
    Zr = Z.clone().requires_grad_(True)
    ld, tr = objective_stochastic_t(Zr, X, state, alpha, model_type, probes, full_set_size, **kw)
    g, = torch.autograd.grad(ld + tr, Zr)
    return float((ld + tr).detach()), g


# ---- gradients w.r.t. Z: ``jax.value_and_grad`` of the objectives (``src/train_inducing.py:195-196``) -------------
def _ggn_dense_t(state, Z, model_type, N):
    """``src/ggn.py:149-193`` kept differentiable in Z (``torch.func.jacrev`` per example)."""
    from torch.func import jacrev
    from .ggn import _model_fun
    flat, unravel = flatten_nn_params(state.params)
    f = _model_fun(state, unravel, model_type)
    M = Z.shape[0]
    G = torch.zeros(flat.numel(), flat.numel(), dtype=flat.dtype)
    for i in range(M):
        J = jacrev(lambda p: f(p, Z[i]).reshape(-1))(flat)
        if model_type == "classifier":
            pr = torch.softmax(f(flat, Z[i]).reshape(-1), dim=-1)
            G = G + J.T @ (torch.diag(pr) - torch.outer(pr, pr)) @ J
        else:
            G = G + J.T @ J
    if model_type == "regressor":
        G = G * torch.exp(-state.params["logvar"]["logvar"]).reshape(())
    return G * (N / M)


def objective_dense_t(Z, X, state, alpha, model_type, full_set_size):
    """``alternative_objective_dense`` (``:176-193``) as a differentiable torch scalar.  ``S`` / ``S_z`` there are what
    ``compute_curvature_approx_dense`` returns — the precisions GGN + alpha I — so the value is
    tr(P inv(P_z)) - log det inv(P_z) = tr(P P_z^-1) + log det P_z."""
    N = full_set_size
    D = flatten_nn_params(state.params)[0].numel()
    I = torch.eye(D, dtype=Z.dtype)
    S = _ggn_dense_t(state, X, model_type, N).detach() + alpha * I
    S_z = _ggn_dense_t(state, Z, model_type, N) + alpha * I
    S_z_inv = torch.linalg.inv(S_z)
    trace_term = torch.trace(S @ S_z_inv)
    logdet_term = -torch.linalg.slogdet(S_z_inv)[1]
    return logdet_term + trace_term


def objective_scalable_t(Z, X, state, alpha, model_type, full_set_size):
    """What ``alternative_objective_scalable`` estimates and ``_exact`` (``:26-84``) evaluates: the dense objective
    minus its two Z-independent terms D and (gamma/alpha) tr(G_X at N/M = 1) (dropped at ``:69`` and ``:80-82``)."""
    N, Kb = full_set_size, X.shape[0]
    D = flatten_nn_params(state.params)[0].numel()
    GX = _ggn_dense_t(state, X, model_type, N).detach() * (Kb / N)          # sum over the batch, no recalibration
    return objective_dense_t(Z, X, state, alpha, model_type, full_set_size) - D - (N / Kb) / alpha * torch.trace(GX)


def variational_grad(objective, Z, X, state, alpha, model_type, full_set_size):
    """(value, d value / dZ) by reverse mode through the dense construction — what ``jax.value_and_grad`` returns."""
    Zr = Z.clone().requires_grad_(True)
    val = objective(Zr, X, state, alpha, model_type, full_set_size)
    g, = torch.autograd.grad(val, Zr)
    return float(val.detach()), g
