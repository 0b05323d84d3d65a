"""Oracle restatement of ``src/sample.py`` (TEST INFRASTRUCTURE — see ``oracle/__init__.py``).

``clip_min`` reproduces the reference's monkey-patched eigenvalue clip
(``src/matfree_monkeypatch.py:19`` via ``src/sample.py:113``): the reference evaluates
f(max(lambda, 1)) with f = x^(-1/2).  ``clip_min=1.0`` is the reference's behaviour
("parity unpinned": no reference test that exercises it can pass);
``clip_min=None`` is the mathematics.  Random draws use torch generators seeded by ``key``
(bit-parity with JAX threefry is not attempted).
"""
from __future__ import annotations

import torch

import lip_amd  # noqa: F401
from lip_amd.utils import flatten_nn_params

from .ggn import build_WTW, compute_W_vps
from .matfree import dense_funm_sym_eigh, funm_lanczos_sym, tridiag_sym

REFERENCE_CLIP_MIN = 1.0
PRODUCT_GRAM_RTOL = 1e-12


def _gram_solver(WTW, gram_rtol):
    """``gram_rtol=None``: the reference's literal ``solve(W^T W, .)`` (``src/sample.py:81-84,135-138``) —
    ill-posed whenever the Gram is (numerically) singular: always for the classifier and for
    neighbouring regression points (cond ~1e17 on the sine fixture).  ``gram_rtol=r``: the truncated
    Moore-Penrose pseudo-inverse the product path uses (eigenvalues <= r * max dropped); identical to
    ``solve`` when the Gram is well conditioned."""
    if gram_rtol is None:
        return lambda u: torch.linalg.solve(WTW, u)
    ev, U = torch.linalg.eigh(WTW)
    inv = torch.where(ev > gram_rtol * ev.max(), 1.0 / ev.clamp_min(1e-300), torch.zeros_like(ev))
    Gp = (U * inv) @ U.T
    return lambda u: Gp @ u


def _normal(key, shape, dtype):
    g = torch.Generator().manual_seed(int(key))
    return torch.randn(shape, generator=g, dtype=dtype)


def inv_matsqrt_dense(state, Z, D, alpha, model_type, full_set_size=None):
    """``src/sample.py:16-52`` (debug twin): materialise W (D, d) and use eigh."""
    flat_params, _ = flatten_nn_params(state.params)
    D = flat_params.shape[0]
    M = Z.shape[0]
    N = full_set_size or M
    beta = N / M
    Wfun, WTfun = compute_W_vps(state, Z, model_type, full_set_size=None)
    I_D = torch.eye(D, dtype=flat_params.dtype)
    W = torch.stack([WTfun(I_D[j]).reshape(-1) for j in range(D)])      # (D, d)   :26
    WT = W.T
    composite = WT @ W
    inv_composite = torch.linalg.solve(composite, torch.eye(composite.shape[0], dtype=W.dtype))
    nullproj = I_D - W @ inv_composite @ WT
    term1 = nullproj / alpha ** 0.5
    I_d = torch.eye(W.shape[1], dtype=W.dtype)
    evals, evecs = torch.linalg.eigh(alpha * I_d + beta * composite)
    inv_sqrt_term = (evecs * (1.0 / torch.sqrt(torch.clamp(evals, min=0.0)))) @ evecs.T
    term2 = W @ inv_composite @ inv_sqrt_term @ WT
    return term1 + term2


def inv_matsqrt_vp(state, Z, D, alpha, model_type, full_set_size=None, key=None, num_proj_steps=1,
                   clip_min=REFERENCE_CLIP_MIN, gram_rtol=None):
    """``src/sample.py:55-145``: v -> W (W^T W)^-1 f(alpha I + beta W^T W) W^T v
    + alpha^(-1/2) (v - W (W^T W)^-1 W^T v), f by 2M-step Lanczos in d-space."""
    Wfun, WTfun = compute_W_vps(state, Z, model_type, full_set_size=None)       # :64
    flat_params, _ = flatten_nn_params(state.params)
    dtype = flat_params.dtype
    dummy = WTfun(torch.zeros(D, dtype=dtype))
    inner_shape = tuple(dummy.shape)
    d = dummy.numel()
    WTW = build_WTW(Wfun, WTfun, inner_shape, d, dtype=dtype, block=2)           # :77
    gsolve = _gram_solver(WTW, gram_rtol)

    def nullproj_vp(v):                                                          # :78-85
        u = WTfun(v).reshape(-1)
        x = gsolve(u)
        return v - Wfun(x.reshape(inner_shape))

    nullproj_term = lambda v: nullproj_vp(v) / alpha ** 0.5                      # :107
    M = Z.shape[0]
    N = full_set_size or M
    beta = N / M
    invsqrt_fun = dense_funm_sym_eigh(lambda x: 1.0 / torch.sqrt(x), clip_min=clip_min)   # :113
    invmatsqrt = funm_lanczos_sym(invsqrt_fun, tridiag_sym(min(2 * M, d)))       # :114-115

    def invmatsqrt_term(V):                                                      # :117-128
        Vflat = V.reshape(-1)
        inner = lambda Uflat: alpha * Uflat + beta * (WTW @ Uflat)
        return invmatsqrt(inner, Vflat).reshape(inner_shape)

    def outer_fun(v):                                                            # :130-139
        u = invmatsqrt_term(WTfun(v)).reshape(-1)
        x = gsolve(u)
        return Wfun(x.reshape(inner_shape))

    return lambda v: outer_fun(v) + nullproj_term(v)                             # :141-143


def sample(state, Z, D, alpha, key, model_type, num_samples=1, full_set_size=None, num_proj_steps=10,
           clip_min=REFERENCE_CLIP_MIN, gram_rtol=None):
    """``src/sample.py:148-156``: zero-mean samples A^(-1/2) eps, eps ~ N(0, I) (S, D)."""
    flat_params, _ = flatten_nn_params(state.params)
    Eps = _normal(key, (num_samples, D), flat_params.dtype)
    fun = inv_matsqrt_vp(state, Z, D, alpha, model_type, full_set_size=full_set_size, key=None,
                         num_proj_steps=num_proj_steps, clip_min=clip_min, gram_rtol=gram_rtol)
    return torch.stack([fun(e) for e in Eps])


def sample_dense(state, Z, D, alpha, key, model_type, num_samples=1, full_set_size=None):
    """``src/sample.py:159-165`` (adds theta_MAP, unlike ``sample``)."""
    A = inv_matsqrt_dense(state, Z, D, alpha, model_type, full_set_size=full_set_size)
    flat_params, _ = flatten_nn_params(state.params)
    Eps = _normal(key, (num_samples, D), flat_params.dtype)
    return Eps @ A.T + flat_params


def sample_both(state, Z, D, alpha, key, model_type, num_samples=1, full_set_size=None,
                clip_min=REFERENCE_CLIP_MIN):
    """``src/sample.py:168-178``."""
    flat_params, _ = flatten_nn_params(state.params)
    Eps = _normal(key, (num_samples, D), flat_params.dtype)
    fun = inv_matsqrt_vp(state, Z, D, alpha, model_type, full_set_size=full_set_size, clip_min=clip_min)
    samples = torch.stack([fun(e) for e in Eps])
    A = inv_matsqrt_dense(state, Z, D, alpha, model_type, full_set_size=full_set_size)
    return samples, Eps @ A.T
