"""Oracle restatement of ``src/ggn.py`` (TEST INFRASTRUCTURE — see ``oracle/__init__.py``).

Literal structure: per-example ``jvp`` -> output Hessian -> ``vjp`` accumulated over the
examples (``src/ggn.py:133-144``); square-root factors W / W^T (``:9-93``); dense GGN
through per-example Jacobians (``:149-193``); column-blocked Gram W^T W (``:198-227``).
One params convention only (SURVEY §4.1-3 is not reproduced): ``state.params`` =
``{'params': {...}[, 'logvar': {...}]}``, ``state.batch_stats`` separate.
"""
from __future__ import annotations

import math

import torch
from torch.func import jacrev, jvp, vjp

import lip_amd  # noqa: F401  (registers the package alias)
from lip_amd.utils import flatten_nn_params


def _model_fun(state, unravel_fn, model_type):
    """``model_fun`` of ``src/ggn.py:42-53,115-123``."""
    def model_fun(pflat, zi):
        p_unr = unravel_fn(pflat)
        if model_type == "regressor":
            return state.apply_fn(p_unr, zi, return_logvar=False)
        variables = dict(p_unr)
        variables["batch_stats"] = state.batch_stats
        return state.apply_fn(variables, zi, train=False, mutable=False)
    return model_fun


def _squeeze(t):
    return t.squeeze()


def compute_W_vps(state, Z, model_type, full_set_size=None, blockwise=False):
    """``src/ggn.py:9-93``."""
    flat_params, unravel_fn = flatten_nn_params(state.params)
    M = Z.shape[0]
    N = full_set_size or M
    recal_term = math.sqrt(N / M)
    model_fun = _model_fun(state, unravel_fn, model_type)

    def sqrt_Hi_apply_T(f_out, vec):      # L . vec   (:16-27)
        if model_type == "regressor":
            c = torch.exp(-state.params["logvar"]["logvar"])
            return torch.sqrt(c) * vec
        p = torch.softmax(f_out, dim=-1)
        s = torch.sqrt(p)
        return s * vec - torch.dot(s, vec) * p

    def sqrt_Hi_apply(f_out, vec):        # L^T . vec (:29-39)
        if model_type == "regressor":
            c = torch.exp(-state.params["logvar"]["logvar"])
            return torch.sqrt(c) * vec
        p = torch.softmax(f_out, dim=-1)
        s = torch.sqrt(p)
        return s * vec - torch.dot(p, vec) * s

    def WT_per_point(i, v):               # :55-62
        zi = Z[i]
        fzi = lambda flatp: _squeeze(model_fun(flatp, zi))
        _, jvp_out = jvp(fzi, (flat_params,), (v,))
        f_val = fzi(flat_params)
        return sqrt_Hi_apply(f_val, jvp_out)

    def W_per_point(i, U_i):              # :64-76
        zi = Z[i]
        fzi = lambda flatp: _squeeze(model_fun(flatp, zi))
        f_val = fzi(flat_params)
        h_sqrt_ui = sqrt_Hi_apply_T(f_val, U_i)
        _, vjp_fn = vjp(fzi, flat_params)
        return vjp_fn(h_sqrt_ui)[0]

    rc_W = lambda i, U_i: recal_term * W_per_point(i, U_i)
    rc_WT = lambda i, v: recal_term * WT_per_point(i, v)
    if blockwise:
        return rc_W, rc_WT

    def WTfun(v):                         # :84-85  -> (M, K) / (M,) for the regressor
        return torch.stack([rc_WT(i, v) for i in range(M)])

    def Wfun(U):                          # :87-91
        per_example = torch.stack([rc_W(i, U[i]) for i in range(M)])
        return per_example.sum(dim=0)

    return Wfun, WTfun


def compute_ggn_vp(state, Z, model_type, full_set_size=None):
    """``src/ggn.py:97-146``: v -> (N/M) sum_i J_i^T H_i J_i v."""
    flat_params, unravel_fn = flatten_nn_params(state.params)
    M = Z.shape[0]
    N = full_set_size or M
    recal_term = N / M
    if model_type == "regressor":
        recal_term = recal_term * torch.exp(-state.params["logvar"]["logvar"])
    model_fun = _model_fun(state, unravel_fn, model_type)

    def H_action(fzi, u):                 # :125-131
        if model_type == "classifier":
            probs = torch.softmax(fzi, dim=-1)
            H_loss = torch.diag(probs) - torch.outer(probs, probs)
            u = H_loss @ u
        return u

    def ggn_vp(v):                        # :133-144
        total = torch.zeros_like(flat_params)
        for i in range(M):
            zi = Z[i]
            fzi = lambda flatp: _squeeze(model_fun(flatp, zi))
            _, jvp_out = jvp(fzi, (flat_params,), (v,))
            f_val = fzi(flat_params)
            hv = H_action(f_val, jvp_out)
            _, vjp_fn = vjp(fzi, flat_params)
            total = total + vjp_fn(hv)[0]
        return total * recal_term

    return ggn_vp


def compute_ggn_vp_batched(state, Z, model_type, full_set_size=None):
    """Same arithmetic as :func:`compute_ggn_vp` with the example loop batched: ONE forward-mode pass and
    ONE reverse-mode pass over all M examples (variant (b) of BASELINE.md §3 — the fair CPU baseline; the
    reference's ``fori_loop`` (``src/ggn.py:144``) is variant (a))."""
    flat_params, unravel_fn = flatten_nn_params(state.params)
    M = Z.shape[0]
    N = full_set_size or M
    recal_term = N / M
    if model_type == "regressor":
        recal_term = recal_term * torch.exp(-state.params["logvar"]["logvar"])
    model_fun = _model_fun(state, unravel_fn, model_type)
    f = lambda flatp: model_fun(flatp, Z).reshape(M, -1)

    def ggn_vp(v):
        out, jv = jvp(f, (flat_params,), (v,))
        if model_type == "classifier":
            probs = torch.softmax(out, dim=-1)
            hv = probs * jv - probs * (probs * jv).sum(-1, keepdim=True)
        else:
            hv = jv
        _, vjp_fn = vjp(f, flat_params)
        return vjp_fn(hv)[0] * recal_term

    return ggn_vp


def compute_ggn_dense(state, Z, model_type, full_set_size=None):
    """``src/ggn.py:149-193``: returns ``(GGN, flat_params, unravel_fn)``."""
    flat_params, unravel_fn = flatten_nn_params(state.params)
    model_fun = _model_fun(state, unravel_fn, model_type)
    M = Z.shape[0]
    D = flat_params.shape[0]
    GGN = torch.zeros(D, D, dtype=flat_params.dtype)
    for i in range(M):
        zi = Z[i]
        J = jacrev(lambda p: model_fun(p, zi))(flat_params)
        J = J.reshape(-1, D)
        if model_type == "classifier":
            probs = torch.softmax(model_fun(flat_params, zi).reshape(-1), dim=-1)
            H_loss = torch.diag(probs) - torch.outer(probs, probs)
            GGN = GGN + J.T @ H_loss @ J
        else:
            GGN = GGN + J.T @ J
    if model_type == "regressor":
        GGN = GGN * torch.exp(-state.params["logvar"]["logvar"])
    N = full_set_size or M
    GGN = GGN * (N / M)
    return GGN, flat_params, unravel_fn


def build_WTW(W, WT, inner_shape, d, *, dtype=torch.float64, block=64):
    """``src/ggn.py:198-227``: dense W^T W one one-hot column at a time, symmetrised
    through its upper triangle (``:227``).  (The block size only bounds memory.)"""
    WTW = torch.zeros(d, d, dtype=dtype)
    for j in range(d):
        e = torch.zeros(d, dtype=dtype)
        e[j] = 1.0
        WTW[:, j] = WT(W(e.reshape(inner_shape))).reshape(-1)
    return torch.triu(WTW) + torch.triu(WTW, 1).T


def build_WTWz(WT, W_z, inner_shape_z, *, d, dtype=torch.float64, block=64):
    """``src/ggn.py:233-272``: cross-Gram W^T W_z (d x d_z)."""
    d_z = math.prod(inner_shape_z)
    G = torch.zeros(d, d_z, dtype=dtype)
    for j in range(d_z):
        e = torch.zeros(d_z, dtype=dtype)
        e[j] = 1.0
        G[:, j] = WT(W_z(e.reshape(inner_shape_z))).reshape(-1)
    return G


def ensure_symmetry(M, jitter=1e-8):
    """``src/ggn.py:277-278``."""
    return 0.5 * (M + M.T) + jitter * torch.eye(M.shape[0], dtype=M.dtype)
