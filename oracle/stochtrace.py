"""Oracle restatement of ``src/stochtrace.py`` (TEST INFRASTRUCTURE — see ``oracle/__init__.py``).

``seed`` is an integer (torch generator seed) where the reference takes a JAX PRNG key.
"""
from __future__ import annotations

import torch

from .matfree import cg


def _rademacher(seed, shape, dtype):
    g = torch.Generator().manual_seed(int(seed))
    return (torch.randint(0, 2, shape, generator=g) * 2 - 1).to(dtype)


def _normal(seed, shape, dtype):
    g = torch.Generator().manual_seed(int(seed))
    return torch.randn(shape, generator=g, dtype=dtype)


def stochastic_trace_estimator_dense(X, seed, num_samples=1_000):
    """``src/stochtrace.py:7-19``."""
    Eps = _rademacher(seed, (num_samples, X.shape[0]), X.dtype)
    return torch.stack([torch.dot(e, X @ e) for e in Eps]).mean()


def stochastic_trace_estimator_mvp(Xfun, D, seed, num_samples=1_000, dtype=torch.float32):
    """``src/stochtrace.py:22-34``."""
    Eps = _rademacher(seed, (num_samples, D), dtype)
    return torch.stack([torch.dot(e, Xfun(e)) for e in Eps]).mean()


def hutchpp_dense(X, seed, num_samples=10):
    """``src/stochtrace.py:37-49``."""
    eps = _normal(seed, (num_samples * 2, X.shape[0]), X.dtype)
    S, G = eps[:num_samples], eps[num_samples:]
    Q, _ = torch.linalg.qr(X @ S.T)
    orthproj = torch.eye(Q.shape[0], dtype=X.dtype) - Q @ Q.T
    return torch.trace(Q.T @ X @ Q) + (1 / num_samples) * torch.trace(G @ orthproj @ X @ orthproj @ G.T)


def hutchpp_mvp(Xfun, D, seed, num_samples=10, dtype=torch.float64):
    """``src/stochtrace.py:52-79``: calls ``Xfun`` on a (D, k) matrix."""
    eps = _normal(seed, (num_samples * 2, D), dtype)
    S, G = eps[:num_samples], eps[num_samples:]
    Q, _ = torch.linalg.qr(Xfun(S.T))
    orthproj = torch.eye(Q.shape[0], dtype=dtype) - Q @ Q.T
    quad_term = lambda M: M.T @ Xfun(M)
    return torch.trace(quad_term(Q)) + (1 / num_samples) * torch.trace(quad_term(orthproj @ G.T))


def hutchpp(Xfun, sampler):
    """``src/stochtrace.py:82-111`` (note the 1/num_samples with num_samples = 2k, ``:89,109``)."""
    eps = sampler(...)
    num_samples = eps.shape[0]
    S, G = eps[: num_samples // 2], eps[num_samples // 2:]
    Q, _ = torch.linalg.qr(torch.stack([Xfun(s) for s in S], dim=1), mode="reduced")
    orthproj = torch.eye(Q.shape[0], dtype=Q.dtype) - Q @ Q.T
    quad_term = lambda M: M.T @ torch.stack([Xfun(M[:, j]) for j in range(M.shape[1])], dim=1)
    return torch.trace(quad_term(Q)) + (1 / num_samples) * torch.trace(quad_term(orthproj @ G.T))


def apply_X(Xfun, M):
    """``src/stochtrace.py:113-114``: rows of M are probes -> (n, k)."""
    return torch.stack([Xfun(m) for m in M], dim=1)


def hutchpp_v2(Xfun, sampler, *, s1, s2):
    """``src/stochtrace.py:118-135``."""
    eps = sampler(...)
    S, G = eps[:s1], eps[s1:]
    Y = apply_X(Xfun, S)
    Q, _ = torch.linalg.qr(Y, mode="reduced")
    XQ = apply_X(Xfun, Q.T)
    low_rank = torch.trace(XQ.T @ Q)
    G_perp = G - (G @ Q) @ Q.T
    XGp = apply_X(Xfun, G_perp)
    resid = torch.trace(G_perp @ XGp) / s2
    return low_rank + resid


def _cg_inverse(Xfun):
    """``Xinvfun`` of ``:144-147``: JAX's CG treats a (D, k) right-hand side as ONE vector
    (all inner products run over every entry), which ``oracle.matfree.cg`` restates."""
    return lambda v: cg(Xfun, v)[0]


def hutchpp_inv_mvp(Xfun, D, seed, num_samples=10, dtype=torch.float64):
    """``src/stochtrace.py:138-148``."""
    return hutchpp_mvp(_cg_inverse(Xfun), D, seed, num_samples=num_samples, dtype=dtype)


def na_hutchpp_dense(X, seed, num_samples=10):
    """``src/stochtrace.py:151-163``."""
    c3 = .25
    eps = _rademacher(seed, (num_samples * 4, X.shape[0]), X.dtype)
    S, R, G = eps[:num_samples], eps[num_samples:3 * num_samples], eps[3 * num_samples:]
    W = X @ S.T
    Z = X @ R.T
    P = torch.linalg.pinv(S @ Z)
    return torch.trace(P @ (W.T @ Z)) + (1 / (c3 * 4 * num_samples)) * (
        torch.trace(G @ X @ G.T) - torch.trace(G @ Z @ P @ W.T @ G.T))


def na_hutchpp_mvp(Xfun, D, seed, num_samples=10, dtype=torch.float32):
    """``src/stochtrace.py:166-180``."""
    c3 = .25
    eps = _rademacher(seed, (num_samples * 4, D), dtype)
    S, R, G = eps[:num_samples], eps[num_samples:3 * num_samples], eps[3 * num_samples:]
    W = Xfun(S.T)
    Z = Xfun(R.T)
    P = torch.linalg.pinv(S @ Z)
    return torch.trace(P @ (W.T @ Z)) + (1 / (c3 * 4 * num_samples)) * (
        torch.trace(G @ Xfun(G.T)) - torch.trace(G @ Z @ P @ W.T @ G.T))


def na_hutchpp_inv_mvp(Xfun, D, seed, num_samples=10):
    """``src/stochtrace.py:183-194`` (casts probes to float64, ``:191``)."""
    inner = _cg_inverse(Xfun)
    return na_hutchpp_mvp(lambda v: inner(v.to(torch.float64)), D, seed, num_samples=num_samples,
                          dtype=torch.float64)
