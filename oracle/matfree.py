"""Restatement of the third-party numerics the reference calls (TEST INFRASTRUCTURE).

* ``matfree`` (un-pinned, ``requirements.txt:5``; source absent from /root/reference):
  ``decomp.tridiag_sym``, ``funm.funm_lanczos_sym``, ``funm.dense_funm_sym_eigh``,
  ``decomp.bidiag``, ``funm.integrand_funm_product_logdet``, ``stochtrace.estimator`` —
  call sites ``src/sample.py:113-115,126``, ``src/train_inducing.py:139-162``,
  ``tests/test_sample.py:337-339``.  Restated from the library's published algorithm:
  k-step Lanczos with full re-orthogonalisation from b/||b||, f(A)b ~= ||b|| Q^T f(T) e1.
* the reference's monkey-patch ``src/matfree_monkeypatch.py:8-22`` (eigenvalue clip >= 1).
* ``jax.scipy.sparse.linalg.cg`` defaults (x0 = 0, tol = 1e-5, atol = 0, maxiter = 10 n) —
  call sites ``src/stochtrace.py:146,192``, ``src/sample.py:71``.
"""
from __future__ import annotations

import torch


def tridiag_sym(num_matvecs: int, reortho: str = "full"):
    """matfree ``decomp.tridiag_sym``: returns ``decompose(matvec, v0) -> (Q, T)`` with
    Q (k, n) orthonormal rows, T (k, k) symmetric tridiagonal, Q A Q^T = T."""

    def decompose(matvec, v0):
        n = v0.numel()
        k = int(num_matvecs)
        if k > n:
            raise ValueError(f"num_matvecs={k} exceeds dimension {n}")
        Q = torch.zeros(k, n, dtype=v0.dtype)
        diag = torch.zeros(k, dtype=v0.dtype)
        off = torch.zeros(max(k - 1, 0), dtype=v0.dtype)
        q = v0 / torch.linalg.vector_norm(v0)
        q_prev = torch.zeros_like(q)
        beta = torch.zeros((), dtype=v0.dtype)
        for j in range(k):
            Q[j] = q
            w = matvec(q)
            w = w - beta * q_prev
            a = torch.dot(q, w)
            w = w - a * q
            if reortho == "full":
                # classical Gram-Schmidt against all previous vectors, twice
                for _ in range(2):
                    w = w - Q[: j + 1].T @ (Q[: j + 1] @ w)
            diag[j] = a
            if j + 1 < k:
                beta = torch.linalg.vector_norm(w)
                if float(beta) <= 1e-13 * max(float(abs(a)), 1e-300):
                    # Krylov space exhausted (matfree has no guard here and would divide by ~0):
                    # decouple the remaining block (diag 1, offdiag 0) so that f(T) e1 is unchanged.
                    diag[j + 1:] = 1.0
                    break
                off[j] = beta
                q_prev = q
                q = w / beta
        T = torch.diag(diag)
        if k > 1:
            T = T + torch.diag(off, 1) + torch.diag(off, -1)
        return Q, T

    return decompose


def dense_funm_sym_eigh(matfun, clip_min=None):
    """matfree ``funm.dense_funm_sym_eigh``; with ``clip_min=1.0`` it is the reference's
    monkey-patched version (``src/matfree_monkeypatch.py:8-22``, clip at ``:19``)."""

    def fun(dense_matrix):
        eigvals, eigvecs = torch.linalg.eigh(dense_matrix)
        if clip_min is not None:
            eigvals = torch.clamp(eigvals, min=clip_min)
        return eigvecs @ torch.diag(matfun(eigvals)) @ eigvecs.T

    return fun


def funm_lanczos_sym(dense_funm, tridiag):
    """matfree ``funm.funm_lanczos_sym``: ``estimate(matvec, vec) ~= f(A) vec``."""

    def estimate(matvec, vec):
        length = torch.linalg.vector_norm(vec)
        Q, T = tridiag(matvec, vec / length)
        fT = dense_funm(T)
        return length * (Q.T @ fT[:, 0])

    return estimate


def bidiag(num_matvecs: int):
    """matfree ``decomp.bidiag`` (Golub-Kahan, full re-orthogonalisation), started in
    the domain: A V = U B, B (k, k) upper bidiagonal.  ``src/train_inducing.py:156``."""

    def decompose(matvec, vecmat, v0):
        k = int(num_matvecs)
        v = v0 / torch.linalg.vector_norm(v0)
        Vs, Us = [], []
        alphas, betas = [], []
        u_prev, beta = None, None
        for j in range(k):
            Vs.append(v)
            u = matvec(v)
            if u_prev is not None:
                u = u - beta * u_prev
            for _ in range(2):
                for uu in Us:
                    u = u - torch.dot(uu, u) * uu
            alpha = torch.linalg.vector_norm(u)
            u = u / alpha
            Us.append(u)
            alphas.append(alpha)
            if j + 1 < k:
                w = vecmat(u) - alpha * v
                for _ in range(2):
                    for vv in Vs:
                        w = w - torch.dot(vv, w) * vv
                beta = torch.linalg.vector_norm(w)
                betas.append(beta)
                v = w / beta
                u_prev = u
        B = torch.diag(torch.stack(alphas))
        if k > 1:
            B = B + torch.diag(torch.stack(betas), 1)
        return torch.stack(Us), B, torch.stack(Vs)

    return decompose


def integrand_funm_product_logdet(bidiag_fn):
    """matfree ``funm.integrand_funm_product_logdet``: v -> ||v||^2 e1^T log(B^T B) e1,
    an unbiased-per-probe quadrature for log det(A^T A).  ``src/train_inducing.py:157``."""

    def quadform(matvec, vecmat, v0):
        length = torch.linalg.vector_norm(v0)
        _, B, _ = bidiag_fn(matvec, vecmat, v0 / length)
        _, S, Vt = torch.linalg.svd(B)
        fx = torch.log(S ** 2)
        return length ** 2 * torch.dot(Vt[:, 0], fx * Vt[:, 0])

    return quadform


def cg(A, b, x0=None, tol=1e-5, atol=0.0, maxiter=None):
    """``jax.scipy.sparse.linalg.cg`` with its defaults; returns ``(x, None)`` like JAX."""
    shape = b.shape
    bf = b.reshape(-1)
    n = bf.numel()
    if maxiter is None:
        maxiter = 10 * n
    Af = lambda v: A(v.reshape(shape)).reshape(-1)
    x = torch.zeros_like(bf) if x0 is None else x0.reshape(-1).clone()
    bs = torch.dot(bf, bf)
    atol2 = max(float(tol) ** 2 * float(bs), float(atol) ** 2)
    r = bf - Af(x)
    p = r.clone()
    gamma = torch.dot(r, r)
    k = 0
    while float(gamma) > atol2 and k < maxiter:
        Ap = Af(p)
        alpha = gamma / torch.dot(p, Ap)
        x = x + alpha * p
        r = r - alpha * Ap
        gamma_new = torch.dot(r, r)
        p = r + (gamma_new / gamma) * p
        gamma = gamma_new
        k += 1
    return x.reshape(shape), None
