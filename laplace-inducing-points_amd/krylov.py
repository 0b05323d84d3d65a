"""Host drivers of the Krylov recurrences over the HIP primitives (``csrc/lip_krylov.hip``).

All vectors are blocks ``(P, N)`` of float32 on the GPU: P independent recurrences advance in lock
step, the small (k x k) / per-probe scalars stay on the device, and the only host synchronisation is
CG's convergence poll.

``lanczos_tridiag`` + ``funm_lanczos_sym`` restate matfree's ``decomp.tridiag_sym`` /
``funm.funm_lanczos_sym`` as the reference calls them (``src/sample.py:113-115,126``):
k-step Lanczos from b/||b|| with full re-orthogonalisation (blocked classical Gram-Schmidt applied
twice = two GEMV passes over Q), f(A) b ~= ||b|| Q^T f(T) e1.  ``cg`` restates
``jax.scipy.sparse.linalg.cg`` with its defaults (``src/stochtrace.py:146,192``, ``src/sample.py:71``).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch

from . import _native as nv


def _chk(X: torch.Tensor) -> torch.Tensor:
    if not (X.is_cuda and X.dtype == torch.float32 and X.is_contiguous()):
        raise nv.NativeError("Krylov primitives take contiguous float32 CUDA blocks (no CPU fallback)")
    return X


def bdot(X: torch.Tensor, Y: torch.Tensor) -> torch.Tensor:
    """out[p] = <X[p], Y[p]>"""
    lib = nv.load()
    P, N = _chk(X).shape
    out = torch.empty(P, device=X.device, dtype=torch.float32)
    nv.check(lib.lip_bdot(nv.ptr(X), nv.ptr(_chk(Y)), nv.ptr(out), P, N, nv.stream_ptr()), "lip_bdot")
    return out


def axpby(Y, X, a: Optional[torch.Tensor] = None, a_s: float = 1.0, b: Optional[torch.Tensor] = None, b_s: float = 1.0):
    """in place: Y[p] = (a_s a[p]) X[p] + (b_s b[p]) Y[p]"""
    lib = nv.load()
    P, N = _chk(Y).shape
    nv.check(lib.lip_axpby(nv.ptr(Y), nv.ptr(_chk(X)), nv.ptr(a), float(a_s), nv.ptr(b), float(b_s), P, N,
                           nv.stream_ptr()), "lip_axpby")
    return Y


def fill_rademacher(P: int, N: int, seed: int, device="cuda") -> torch.Tensor:
    lib = nv.load()
    X = torch.empty(P, N, device=device, dtype=torch.float32)
    nv.check(lib.lip_fill_rademacher(nv.ptr(X), P, N, int(seed) & (2 ** 64 - 1), nv.stream_ptr()), "lip_fill_rademacher")
    return X


def fill_normal(P: int, N: int, seed: int, device="cuda", out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = nv.load()
    X = torch.empty(P, N, device=device, dtype=torch.float32) if out is None else _chk(out)
    assert tuple(X.shape) == (P, N)
    nv.check(lib.lip_fill_normal(nv.ptr(X), P, N, int(seed) & (2 ** 64 - 1), nv.stream_ptr()), "lip_fill_normal")
    return X


def _rows_f32(X: torch.Tensor):
    """(rows, N) float32 device matrix whose rows are contiguous (any row stride): (tensor, row stride)."""
    if not (X.is_cuda and X.dtype == torch.float32 and X.dim() == 2 and X.stride(1) == 1):
        raise nv.NativeError("expected a float32 CUDA matrix with contiguous rows (no CPU fallback)")
    return X, (X.stride(0) if X.shape[0] > 1 else X.shape[1])


def dot_nt(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """A B^T (m, n) in float64 from float32 rows A (m, K), B (n, K): every product exact, float64 accumulation
    (``lip_dot_nt_f64``)."""
    lib = nv.load()
    A, lda = _rows_f32(A)
    B, ldb = _rows_f32(B)
    if A.shape[1] != B.shape[1]:
        raise ValueError(f"dot_nt: inner dimensions differ ({A.shape[1]} vs {B.shape[1]})")
    C = torch.empty(A.shape[0], B.shape[0], device=A.device, dtype=torch.float64)
    nv.check(lib.lip_dot_nt_f64(A.data_ptr(), lda, A.shape[0], B.data_ptr(), ldb, B.shape[0], A.shape[1], C.data_ptr(),
                                nv.stream_ptr()), "lip_dot_nt_f64")
    return C


def gemm_nt(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """A B^T (m, n) float32 for float32 rows A (m, K), B (n, K), K >> m, n: the hand-written split-K MFMA kernel
    (``lip_gemm_nt``)."""
    lib = nv.load()
    A, lda = _rows_f32(A)
    B, ldb = _rows_f32(B)
    if A.shape[1] != B.shape[1]:
        raise ValueError(f"gemm_nt: inner dimensions differ ({A.shape[1]} vs {B.shape[1]})")
    C = torch.empty(A.shape[0], B.shape[0], device=A.device, dtype=torch.float32)
    nv.check(lib.lip_gemm_nt(A.data_ptr(), lda, A.shape[0], B.data_ptr(), ldb, B.shape[0], A.shape[1], C.data_ptr(),
                             nv.stream_ptr()), "lip_gemm_nt")
    return C


def gemm_nn_axpy(T: torch.Tensor, B: torch.Tensor, V: Optional[torch.Tensor] = None, beta: float = 1.0,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """T B + beta V for T (m, k) float32, B (k, N) float32 rows, V (m, N) or None; ``out`` may be V (in place).  The
    hand-written NN-layout MFMA kernel with the addend in its epilogue (``lip_gemm_nn_axpy``)."""
    lib = nv.load()
    T, ldt = _rows_f32(T)
    B, ldb = _rows_f32(B)
    if T.shape[1] != B.shape[0]:
        raise ValueError(f"gemm_nn_axpy: inner dimensions differ ({T.shape[1]} vs {B.shape[0]})")
    m, N = T.shape[0], B.shape[1]
    if T.shape[1] < 4 or N < 4:                     # below the kernel's vector width: a handful of rows / columns
        res = T @ B if V is None else torch.addmm(V, T, B, beta=float(beta))
        return res if out is None else out.copy_(res)
    O = torch.empty(m, N, device=B.device, dtype=torch.float32) if out is None else out
    O, ldo = _rows_f32(O)
    vp, ldv = 0, 0
    if V is not None:
        V, ldv = _rows_f32(V)
        vp = V.data_ptr()
    nv.check(lib.lip_gemm_nn_axpy(T.data_ptr(), ldt, m, T.shape[1], B.data_ptr(), ldb, N, vp, ldv, float(beta), O.data_ptr(), ldo,
                                  nv.stream_ptr()), "lip_gemm_nn_axpy")
    return O


def rows_combine(Cm: torch.Tensor, Y: torch.Tensor, Z: Optional[torch.Tensor] = None, zscale: float = 0.0,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[i] = zscale * Z[i] + sum_j Cm[i, j] Y[j]: r combinations of the s rows of Y in one streaming pass
    (``lip_rows_combine``).  Cm (r, s) is taken in float64."""
    lib = nv.load()
    Y, ldy = _rows_f32(Y)
    r, s = Cm.shape
    if s != Y.shape[0]:
        raise ValueError(f"rows_combine: {s} coefficients per row for {Y.shape[0]} rows")
    Cd = Cm.to(device=Y.device, dtype=torch.float64).contiguous()
    N = Y.shape[1]
    O = torch.empty(r, N, device=Y.device, dtype=torch.float32) if out is None else out
    O, ldo = _rows_f32(O)
    zp, ldz = 0, 0
    if Z is not None:
        Z, ldz = _rows_f32(Z)
        zp = Z.data_ptr()
    nv.check(lib.lip_rows_combine(Cd.data_ptr(), Y.data_ptr(), ldy, s, zp, ldz, float(zscale), O.data_ptr(), ldo, r, N,
                                  nv.stream_ptr()), "lip_rows_combine")
    return O


def gram_orthonormalize(Y: torch.Tensor, rtol: float = 1e-10, passes: int = 2, return_transform: bool = False):
    """Orthonormal rows spanning the rows of Y (s, N), N >> s, without a Householder QR of a tall matrix: the float64
    Gram G = Y Y^T (``dot_nt``: one read of Y), its s x s eigendecomposition G = U L U^T, Q = L^(-1/2) U^T Y
    (``rows_combine``: one read + one write) — CholeskyQR with the Cholesky factor where it exists and the symmetric
    factor L^(1/2) U^T otherwise, so a rank-deficient Y (Hutch++ with more probes than dimensions,
    ``tests/test_stochtrace.py:90-97``) simply yields fewer rows.  Done twice ("CholeskyQR2"): the first pass leaves ||Q Q^T - I|| ~ eps cond(Y)^2,
    the second brings it to rounding.  12 N s bytes per pass (>= 3 * 4 * D * s).  Replaces
    ``jnp.linalg.qr`` at ``src/stochtrace.py:128`` (only span(Q) enters the estimator).  ``return_transform`` also
    returns the (r, s) float64 matrix T with Q = T Y (the adjoint of the orthonormalisation needs it,
    ``stochastic_grad.py``)."""
    s_rows, N = Y.shape
    if s_rows == 0 or not bool((Y != 0).any()):
        # rank 0 (a zero operator output): the reference's QR simply proceeds with nothing to deflate
        Q0 = torch.zeros(0, N, device=Y.device, dtype=torch.float32)
        return (Q0, torch.zeros(0, s_rows, device=Y.device, dtype=torch.float64)) if return_transform else Q0
    if s_rows > 384 or N < 4 * s_rows:
        # not tall-skinny (the reference's own "more probes than dimensions" test): Householder QR in float64
        Qf, Rf = torch.linalg.qr(Y.T.double(), mode="reduced")
        if return_transform:                        # Q = R^-T Y (full column rank assumed on this path)
            Tm = torch.linalg.solve_triangular(Rf.T, torch.eye(Rf.shape[0], device=Y.device, dtype=torch.float64), upper=False)
            return Qf.T.float().contiguous(), Tm
        return Qf.T.float().contiguous()
    Q = Y
    Ttot = None                                     # running transform: Q = Ttot Y
    for it in range(passes):
        G = dot_nt(Q, Q)
        G = 0.5 * (G + G.T)
        Cm = None
        if s_rows <= 64:
            # CholeskyQR proper: Q = L^-1 Y with G = L L^T (an s x s potrf + trtri is several times cheaper than the
            # symmetric eigensolver at these sizes); a failed factorisation (rank-deficient or very ill-conditioned
            # Y) falls through to the eigen-decomposition, which drops the null directions
            L, info = torch.linalg.cholesky_ex(G)
            if int(info.item()) == 0:
                dg = L.diagonal()
                if float(dg.min() / dg.max()) > (1e-5 if it == 0 else 0.25):
                    Cm = torch.linalg.solve_triangular(L, torch.eye(Q.shape[0], device=G.device, dtype=G.dtype), upper=False)
        if Cm is None:
            ev, U = torch.linalg.eigh(G)
            keep = ev > (rtol if it == 0 else 0.25) * ev.max().clamp_min(1e-300)
            Cm = (U[:, keep] * torch.rsqrt(ev[keep])).T
        Q = rows_combine(Cm, Q)
        Ttot = Cm if Ttot is None else Cm @ Ttot
    return (Q, Ttot) if return_transform else Q


def lanczos_tridiag(matvec: Callable[[torch.Tensor], torch.Tensor], V0: torch.Tensor, k: int
                    ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """k-step Lanczos with full re-orthogonalisation, P recurrences at once.

    ``matvec`` maps a (P, N) block to a (P, N) block.  Returns Q (P, k, N) with orthonormal rows,
    diag (P, k), offdiag (P, k-1).  Bytes per step j (fp32): (7 + 4 (j+1)) N x 4 per probe."""
    lib = nv.load()
    P, N = _chk(V0).shape
    if k > N:
        raise ValueError(f"num_matvecs={k} exceeds dimension {N}")
    st = nv.stream_ptr()
    dev = V0.device
    ldq = (N + 3) // 4 * 4                     # padded row stride: aligned float4 streaming of the basis
    Qbuf = torch.empty(P, k, ldq, device=dev, dtype=torch.float32)
    Q = Qbuf[:, :, :N]
    diag = torch.zeros(P, k, device=dev, dtype=torch.float32)
    off = torch.zeros(P, max(k - 1, 0), device=dev, dtype=torch.float32)
    c1 = torch.empty(P, k, device=dev, dtype=torch.float32)
    c2 = torch.empty(P, k, device=dev, dtype=torch.float32)
    nrm2 = bdot(V0, V0)
    nv.check(lib.lip_scale_store(nv.ptr(V0), nv.ptr(nrm2), nv.ptr(Qbuf), 0, P, k, N, ldq, st), "lip_scale_store")
    alive = torch.ones(P, device=dev, dtype=torch.bool)
    for j in range(k):
        w = matvec(Q[:, j].contiguous()).contiguous()      # (P, N) copy of basis row j (N may be < ldq)
        _chk(w)
        nv.check(lib.lip_multi_dot(nv.ptr(Qbuf), nv.ptr(w), nv.ptr(c1), P, j + 1, k, N, ldq, st), "lip_multi_dot")
        nv.check(lib.lip_multi_axpy_norm(nv.ptr(Qbuf), nv.ptr(c1), nv.ptr(w), nv.ptr(nrm2), P, j + 1, k, N, ldq, st), "lip_multi_axpy_norm")
        nv.check(lib.lip_multi_dot(nv.ptr(Qbuf), nv.ptr(w), nv.ptr(c2), P, j + 1, k, N, ldq, st), "lip_multi_dot")
        nv.check(lib.lip_multi_axpy_norm(nv.ptr(Qbuf), nv.ptr(c2), nv.ptr(w), nv.ptr(nrm2), P, j + 1, k, N, ldq, st), "lip_multi_axpy_norm")
        # breakdown guard: once the Krylov space of a probe is exhausted (residual at rounding level)
        # the remaining basis vectors are zero and the tridiagonal block decouples (diag 1, offdiag 0),
        # which leaves f(T) e1 unchanged instead of producing 0/0.
        diag[:, j] = torch.where(alive, c1[:, j] + c2[:, j], torch.ones_like(nrm2))
        if j + 1 < k:
            # ||A q_j||^2 without a pass over w: the basis is orthonormal, so it is the residual's norm plus the
            # squares of the coefficients taken out (Pythagoras; only a 1e-10-relative breakdown threshold hangs on it)
            wn2 = nrm2 + ((c1[:, :j + 1] + c2[:, :j + 1]) ** 2).sum(1)
            alive = alive & (nrm2 > (1e-10 * wn2).clamp_min(1e-36))
            off[:, j] = torch.where(alive, torch.sqrt(nrm2), torch.zeros_like(nrm2))
            safe = torch.where(alive, nrm2, torch.full_like(nrm2, float("inf")))
            nv.check(lib.lip_scale_store(nv.ptr(w), nv.ptr(safe), nv.ptr(Qbuf), j + 1, P, k, N, ldq, st), "lip_scale_store")
    Q.basis_buffer, Q.ldq = Qbuf, ldq      # the padded storage the HIP kernels take
    return Q, diag, off


def tridiag_dense(diag: torch.Tensor, off: torch.Tensor) -> torch.Tensor:
    T = torch.diag_embed(diag)
    if off.shape[-1] > 0:
        T = T + torch.diag_embed(off, 1) + torch.diag_embed(off, -1)
    return T


def dense_funm_sym_eigh(matfun: Callable, clip_min: Optional[float] = None, floor: Optional[float] = None):
    """Dense f(T) through eigh; ``clip_min=1.0`` reproduces the reference's monkey-patch
    (``src/matfree_monkeypatch.py:8-22``, clip at ``:19``).  ``floor`` is a known lower bound of the
    operator's spectrum (alpha for alpha I + PSD): fp32 Ritz values that rounding pushed below it are
    raised to it instead of producing NaN under x^(-1/2).  Batched over leading dimensions."""
    def fun(T):
        ev, U = torch.linalg.eigh(T)
        if floor is not None:
            ev = torch.clamp(ev, min=floor)
        if clip_min is not None:
            ev = torch.clamp(ev, min=clip_min)
        return (U * matfun(ev).unsqueeze(-2)) @ U.transpose(-1, -2)
    return fun


def funm_lanczos_sym(dense_funm: Callable, num_matvecs: int):
    """``estimate(matvec, B)`` ~= f(A) B[p] for every row of the block B (P, N)."""
    lib = nv.load()

    def estimate(matvec, B):
        B = _chk(B.contiguous())
        P, N = B.shape
        k = int(num_matvecs)
        length = torch.sqrt(bdot(B, B))
        Q, diag, off = lanczos_tridiag(matvec, B, k)
        fT = dense_funm(tridiag_dense(diag.double(), off.double()))
        coef = (-(fT[:, :, 0] * length.double()[:, None])).float().contiguous()      # (P, k)
        out = torch.zeros(P, N, device=B.device, dtype=torch.float32)
        nrm = torch.empty(P, device=B.device, dtype=torch.float32)
        nv.check(lib.lip_multi_axpy_norm(nv.ptr(Q.basis_buffer), nv.ptr(coef), nv.ptr(out), nv.ptr(nrm), P, k, k, N, Q.ldq,
                                         nv.stream_ptr()), "lip_multi_axpy_norm")
        return out

    return estimate


def funm_lanczos_dense(dense_funm: Callable, num_matvecs: int):
    """``estimate(A, B)`` ~= f(A) B[p] for the rows of B (S, d) and an EXPLICIT symmetric A (d, d), all float64 on the
    device: the reference's small-space Lanczos (``src/sample.py:113-128`` runs matfree's ``funm_lanczos_sym`` on the
    dense d x d matrix alpha I + beta W^T W, d = M K <= ~1000).  Same recurrence as :func:`lanczos_tridiag` — full
    re-orthogonalisation (two Gram-Schmidt passes), breakdown guard — as batched torch algebra: at d ~ 500 the vectors
    are three orders of magnitude shorter than what the streaming HIP kernels are built for, and the matrix spans nine
    decades, so the recurrence runs in float64."""
    def estimate(A, B):
        S, d = B.shape
        k = min(int(num_matvecs), d)
        length = B.norm(dim=1)
        Q = torch.zeros(S, k, d, device=B.device, dtype=B.dtype)
        diag = torch.ones(S, k, device=B.device, dtype=B.dtype)
        off = torch.zeros(S, max(k - 1, 0), device=B.device, dtype=B.dtype)
        q = B / length.clamp_min(1e-300)[:, None]
        alive = length > 0
        for j in range(k):
            Q[:, j] = q
            w = q @ A
            wn = w.norm(dim=1)
            Qj = Q[:, :j + 1]
            c1 = torch.einsum("skd,sd->sk", Qj, w)
            w = w - torch.einsum("sk,skd->sd", c1, Qj)
            c2 = torch.einsum("skd,sd->sk", Qj, w)
            w = w - torch.einsum("sk,skd->sd", c2, Qj)
            diag[:, j] = torch.where(alive, c1[:, j] + c2[:, j], torch.ones_like(wn))
            if j + 1 < k:
                nrm = w.norm(dim=1)
                alive = alive & (nrm > 1e-13 * wn.clamp_min(1e-300))
                off[:, j] = torch.where(alive, nrm, torch.zeros_like(nrm))
                q = torch.where(alive[:, None], w / nrm.clamp_min(1e-300)[:, None], torch.zeros_like(w))
        fT = dense_funm(tridiag_dense(diag, off))
        return torch.einsum("sk,skd->sd", fT[:, :, 0] * length[:, None], Q)
    return estimate


def cg(A: Callable[[torch.Tensor], torch.Tensor], B: torch.Tensor, x0: Optional[torch.Tensor] = None, tol: float = 1e-5,
       atol: float = 0.0, maxiter: Optional[int] = None, check_every: int = 1, stall: Optional[int] = None,
       keep_best: bool = False):
    """Batched conjugate gradients, one independent solve per row of B (P, N), with JAX's defaults and
    stopping rule (||r||^2 <= max(tol^2 ||b||^2, atol^2), maxiter = 10 N).  Returns ``(X, info)`` where
    info holds the iteration count and final residual norms (the reference discards it).

    ``stall`` (not in JAX): a right-hand side whose recurrence residual has not dropped by 10 % below its best value for
    ``stall`` consecutive iterations is frozen — a float32 recurrence that has reached the noise floor of its operator
    only accumulates rounding when iterated further (forward error 8e-3 after 5 steps, 8e-2 after 200 at the CIFAR
    config's alpha = 0.005, deflated operator).

    ``keep_best`` (not in JAX): after every step the TRUE residual b - A x is evaluated (one more product per iteration)
    and, per right-hand side, the iterate with the smallest one is what is returned; a right-hand side stops when its
    true residual meets the tolerance or has not improved for ``stall`` (default 2) steps.  For operators whose float32
    product carries noise of the size of the tolerance: the recurrence residual then says nothing about x, and a step
    taken along a noise-dominated direction can throw the iterate far off (seen: forward error 2e-1 after 4 steps where
    the first iterate was at 1e-3)."""
    lib = nv.load()
    B = _chk(B.contiguous())
    P, N = B.shape
    st = nv.stream_ptr()
    if maxiter is None:
        maxiter = 10 * N
    X = torch.zeros_like(B) if x0 is None else _chk(x0.clone().contiguous())
    R = B.clone() if x0 is None else (B - A(X)).contiguous()
    Pd = R.clone()
    rr = bdot(R, R)
    atol2 = torch.clamp(tol * tol * bdot(B, B), min=atol * atol)
    active = (rr > atol2).to(torch.int32)
    rr_new = torch.empty_like(rr)
    best, since = rr.clone(), torch.zeros_like(active)
    it = 0
    if keep_best:
        patience = 2 if stall is None else int(stall)
        Xb = X.clone()
        tb = bdot(R, R)                                  # true residual of the iterate kept (x0: r = b - A x0)
        alive = (tb > atol2)
        idle = torch.zeros_like(active)
        while it < maxiter and bool(alive.any()):
            act32 = alive.to(torch.int32)
            Ap = _chk(A(Pd).contiguous())
            pAp = bdot(Pd, Ap)
            nv.check(lib.lip_cg_update(nv.ptr(X), nv.ptr(R), nv.ptr(Pd), nv.ptr(Ap), nv.ptr(rr), nv.ptr(pAp), nv.ptr(act32),
                                       nv.ptr(rr_new), P, N, st), "lip_cg_update")
            nv.check(lib.lip_cg_direction(nv.ptr(Pd), nv.ptr(R), nv.ptr(rr_new), nv.ptr(rr), nv.ptr(act32), P, N, st),
                     "lip_cg_direction")
            rr = torch.where(alive, rr_new, rr)
            Rt = (B - _chk(A(X).contiguous()))
            tt = bdot(Rt, Rt)
            better = alive & (tt < tb)
            Xb = torch.where(better[:, None], X, Xb)
            idle = torch.where(tt < 0.81 * tb, torch.zeros_like(idle), idle + 1)
            tb = torch.where(better, tt, tb)
            alive = alive & (tb > atol2) & (idle < patience)
            it += 1
        return Xb, dict(iterations=it, residual_norm=torch.sqrt(tb))
    while it < maxiter:
        if it % check_every == 0 and not bool(active.any()):
            break
        Ap = _chk(A(Pd).contiguous())
        pAp = bdot(Pd, Ap)
        nv.check(lib.lip_cg_update(nv.ptr(X), nv.ptr(R), nv.ptr(Pd), nv.ptr(Ap), nv.ptr(rr), nv.ptr(pAp), nv.ptr(active),
                                   nv.ptr(rr_new), P, N, st), "lip_cg_update")
        nv.check(lib.lip_cg_direction(nv.ptr(Pd), nv.ptr(R), nv.ptr(rr_new), nv.ptr(rr), nv.ptr(active), P, N, st),
                 "lip_cg_direction")
        rr = torch.where(active.bool(), rr_new, rr)
        active = (rr > atol2).to(torch.int32) * active
        if stall is not None:
            since = torch.where(rr < 0.81 * best, torch.zeros_like(since), since + 1)
            best = torch.minimum(best, rr)
            active = active * (since < int(stall)).to(torch.int32)
        it += 1
    return X, dict(iterations=it, residual_norm=torch.sqrt(rr))


def cg_dense(A: torch.Tensor, B: torch.Tensor, tol: float = 1e-5, atol: float = 0.0, maxiter: Optional[int] = None):
    """:func:`cg` for an EXPLICIT small symmetric matrix A (d, d) and right-hand sides B (P, d) in float64 on the
    device — the small-space solves of the reference (``src/sample.py:71`` runs JAX's CG on the d x d Gram): same
    stopping rule and defaults as :func:`cg`, plain torch algebra (d is ~1e1..1e3, and these Grams have condition
    numbers far beyond what a float32 recurrence resolves)."""
    A = A.double()
    B = B.double()
    P, d = B.shape
    maxiter = 10 * d if maxiter is None else maxiter
    X = torch.zeros_like(B)
    R = B.clone()
    Pd = R.clone()
    rr = (R * R).sum(1)
    atol2 = torch.clamp(tol * tol * (B * B).sum(1), min=atol * atol)
    it = 0
    while it < maxiter and bool((rr > atol2).any()):
        act = rr > atol2
        Ap = Pd @ A
        a = torch.where(act, rr / (Pd * Ap).sum(1).clamp_min(1e-300), torch.zeros_like(rr))
        X = X + a[:, None] * Pd
        R = R - a[:, None] * Ap
        rr_new = torch.where(act, (R * R).sum(1), rr)
        Pd = torch.where(act[:, None], R + (rr_new / rr.clamp_min(1e-300))[:, None] * Pd, Pd)
        rr = rr_new
        it += 1
    return X, dict(iterations=it, residual_norm=torch.sqrt(rr))


def _basis(P: int, k: int, N: int, dev):
    ldq = (N + 3) // 4 * 4
    return torch.empty(P, k, ldq, device=dev, dtype=torch.float32), ldq


def _cgs2(lib, Qbuf, ldq, w, P, j, k, N, c1, c2, nrm2, st):
    """w -= Q_j^T (Q_j w), twice; returns with nrm2 = ||w||^2 (blocked classical Gram-Schmidt, 2 passes)."""
    for c in (c1, c2):
        nv.check(lib.lip_multi_dot(nv.ptr(Qbuf), nv.ptr(w), nv.ptr(c), P, j, k, N, ldq, st), "lip_multi_dot")
        nv.check(lib.lip_multi_axpy_norm(nv.ptr(Qbuf), nv.ptr(c), nv.ptr(w), nv.ptr(nrm2), P, j, k, N, ldq, st),
                 "lip_multi_axpy_norm")


def bidiag(matvec: Callable, vecmat: Callable, V0: torch.Tensor, k: int, n_out: int, return_bases: bool = False):
    """k-step Golub-Kahan bidiagonalisation with full re-orthogonalisation, P recurrences at once, started in
    the domain: A V = U B with B (k, k) upper bidiagonal (matfree ``decomp.bidiag`` as the reference calls it,
    ``src/train_inducing.py:156``).  ``matvec``: (P, N) -> (P, n_out); ``vecmat``: (P, n_out) -> (P, N).
    Returns (alphas (P, k), betas (P, k-1)) and, with ``return_bases``, the bases V (P, k, N) and U (P, k, n_out)
    (views of the padded storage) and the projection coefficients of the two Gram-Schmidt passes of every step
    ((cu1, cu2, cv1, cv2), each (P, k, k)) — what the adjoint recurrence of ``stochastic_grad.py`` walks back over."""
    lib = nv.load()
    P, N = _chk(V0).shape
    st = nv.stream_ptr()
    dev = V0.device
    Vb, ldv = _basis(P, k, N, dev)
    Ub, ldu = _basis(P, k, n_out, dev)
    alphas = torch.zeros(P, k, device=dev, dtype=torch.float32)
    betas = torch.zeros(P, max(k - 1, 0), device=dev, dtype=torch.float32)
    c1 = torch.empty(P, k, device=dev, dtype=torch.float32)
    c2 = torch.empty(P, k, device=dev, dtype=torch.float32)
    coef = [torch.zeros(P, k, k, device=dev, dtype=torch.float32) for _ in range(4)] if return_bases else None   # cu1, cu2, cv1, cv2
    nrm2 = bdot(V0, V0)
    nv.check(lib.lip_scale_store(nv.ptr(V0), nv.ptr(nrm2), nv.ptr(Vb), 0, P, k, N, ldv, st), "lip_scale_store")
    for j in range(k):
        v = Vb[:, j, :N].contiguous()
        u = _chk(matvec(v).contiguous())
        if j > 0:
            _cgs2(lib, Ub, ldu, u, P, j, k, n_out, c1, c2, nrm2, st)
            if coef is not None:
                coef[0][:, j, :j], coef[1][:, j, :j] = c1[:, :j], c2[:, :j]
        else:
            nrm2 = bdot(u, u)
        alphas[:, j] = torch.sqrt(nrm2)
        nv.check(lib.lip_scale_store(nv.ptr(u), nv.ptr(nrm2), nv.ptr(Ub), j, P, k, n_out, ldu, st), "lip_scale_store")
        if j + 1 < k:
            w = _chk(vecmat(Ub[:, j, :n_out].contiguous()).contiguous())
            _cgs2(lib, Vb, ldv, w, P, j + 1, k, N, c1, c2, nrm2, st)
            if coef is not None:
                coef[2][:, j, :j + 1], coef[3][:, j, :j + 1] = c1[:, :j + 1], c2[:, :j + 1]
            betas[:, j] = torch.sqrt(nrm2)
            nv.check(lib.lip_scale_store(nv.ptr(w), nv.ptr(nrm2), nv.ptr(Vb), j + 1, P, k, N, ldv, st), "lip_scale_store")
    if return_bases:
        return alphas, betas, Vb[:, :, :N], Ub[:, :, :n_out], tuple(coef)
    return alphas, betas


def slq_logdet_product(matvec: Callable, vecmat: Callable, V0: torch.Tensor, k: int, n_out: int) -> torch.Tensor:
    """matfree ``funm.integrand_funm_product_logdet(bidiag(k))``: per probe ||v||^2 e1^T log(B^T B) e1, an
    unbiased-per-probe quadrature of log det(A^T A) (``src/train_inducing.py:156-162``).  Returns (P,) float64."""
    length2 = bdot(V0, V0).double()
    alphas, betas = bidiag(matvec, vecmat, V0, k, n_out)
    B = torch.diag_embed(alphas.double())
    if betas.shape[-1] > 0:
        B = B + torch.diag_embed(betas.double(), 1)
    _, S, Vt = torch.linalg.svd(B)
    fx = torch.log(S ** 2)
    return length2 * (Vt[:, :, 0] ** 2 * fx).sum(-1)


class RangeDeflation:
    """Exact treatment of an invariant subspace of a symmetric operator A: orthonormal rows ``Q`` (r, N) with
    A q_k = ``lam[k]`` q_k.  A Krylov recurrence (Lanczos f(A) b, CG) is then run on the complement only — vectors and
    operator outputs projected off range(Q) — and the range part of the answer is added in closed form:

        f(A) b = sum_k q_k f(lam_k) <q_k, b>  +  f(A_perp) b_perp .

    Why this matters in float32: the computed product A v = alpha v + beta W (W^T v) carries rounding noise of size
    eps ||beta W W^T|| ||v||, and that noise lies (to first order) IN range(W) — W maps into it.  At the CIFAR config
    ||beta W W^T|| = 1.5e7 and alpha = 0.005, so the noise (~1 per unit vector) swamps the bottom of the spectrum the
    function x^(-1/2) is steepest on, and more Lanczos steps make it worse (k = 100: 26 % off, ``tests/
    test_sampler_fullsize.py``).  Projected off range(W) the noise goes with it: what is left of the operator is
    alpha I up to eps^2 ||A||.  The subspace comes from the sampler's orthonormalised factor (``sample.py``)."""

    def __init__(self, Q: torch.Tensor, lam: torch.Tensor):
        self.Q, self.lam = _chk(Q.contiguous()), lam.double()

    def coeffs(self, V: torch.Tensor) -> torch.Tensor:
        """<q_k, v> for the rows of V, float64 (S, r)."""
        V = _chk(V.contiguous())
        return dot_nt(V, self.Q) if V.shape[0] < 32 else gemm_nt(V, self.Q).double()

    def project_out(self, V: torch.Tensor, C: Optional[torch.Tensor] = None, passes: int = 1) -> torch.Tensor:
        """V - Q^T (Q V): the rows of V projected off range(Q).  ``passes=2`` repeats the projection on the result: the
        range component of a float32 product is ~200x the complement's signal at alpha = 0.005, and one pass with
        float32 rows q_k (orthonormal to ~1e-7) leaves ~1e-6 of it behind."""
        V = _chk(V.contiguous())
        out = rows_combine(-(self.coeffs(V) if C is None else C), self.Q, Z=V, zscale=1.0)
        for _ in range(passes - 1):
            out = rows_combine(-self.coeffs(out), self.Q, Z=out, zscale=1.0)
        return out

    def wrap(self, matvec: Callable) -> Callable:
        """the operator restricted to the complement: v -> P_perp A v (for v already in the complement).  The product
        runs on the DIRECT implicit GEMMs (Winograd route off for the call): on the complement A v cancels to
        alpha v ~ 1e-9 ||A|| ||v||, the recurrences stop at the product's absolute rounding error, and the direct
        kernels' is ~4x smaller than the Winograd transforms' (deflated CG at the config's alpha: 2 iterations against
        9 for the same answer, tests/test_sampler_fullsize.py); at the 8 - 16 right-hand sides of these solves the
        route is worth 10 % of a product."""
        def complement_product(V):
            with nv.winograd_route(0):
                AV = matvec(V)
            return self.project_out(AV, passes=2)
        return complement_product

    def relative_residual(self, A: Callable, X: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
        """||A x - b|| / ||b|| per row, evaluated in the two invariant subspaces: lam_k <q_k, x> - <q_k, b> on range(Q)
        (float64) and P_perp (A x) - P_perp b on the complement.  Evaluated naively through the float32 product the
        residual of an accurate solution is swamped by the same range(W) noise the deflation removes
        (eps ||A|| ||x|| ~ 300 ||b|| at alpha = 0.005)."""
        rr = self.coeffs(X) * self.lam[None, :] - self.coeffs(B)
        with nv.winograd_route(0):                   # (as in :meth:`wrap`)
            AX = A(X)
        rp = self.project_out(AX, passes=2) - self.project_out(B)
        return torch.sqrt((rr * rr).sum(1) + (rp.double() ** 2).sum(1)) / B.double().norm(dim=1)

    def closed_form(self, B: torch.Tensor, f: Callable, lam_perp: float) -> torch.Tensor:
        """f(A) B when A is lam_perp * I on the complement of range(Q) (A = alpha I + beta W W^T: lam_perp = alpha):
        sum_k q_k f(lam_k) <q_k, b> + f(lam_perp) b_perp — the answer the deflated recurrences converge to, used as their
        reference (a float32-stored x cannot make ||A x - b|| small at cond 3e9: eps cond ~ 180)."""
        C = self.coeffs(B)
        lp = torch.tensor([float(lam_perp)], dtype=torch.float64, device=B.device)
        return axpby(self.range_part(C, f), self.project_out(B, C), None, float(f(lp)[0]), None, 1.0)

    def range_part(self, C: torch.Tensor, f: Callable) -> torch.Tensor:
        """sum_k q_k f(lam_k) C[:, k] -> (S, N)"""
        return rows_combine(C * f(self.lam)[None, :], self.Q)


def cg_deflated(A: Callable, B: torch.Tensor, defl: RangeDeflation, **cg_kw):
    """A^-1 B with the invariant subspace of ``defl`` solved exactly and CG run on the complement (:func:`cg`'s
    arguments and return value; the iteration count is that of the complement solve; ``keep_best`` defaults to True:
    the iterate with the smallest true residual on the complement is returned)."""
    B = _chk(B.contiguous())
    C = defl.coeffs(B)
    cg_kw.setdefault("keep_best", True)             # the complement's product is noise-limited: see :func:`cg`
    Xp, info = cg(defl.wrap(A), defl.project_out(B, C), **cg_kw)
    Xp = defl.project_out(Xp)                       # what the recurrence let leak back into range(Q)
    X = axpby(defl.range_part(C, lambda lam: 1.0 / lam), Xp, None, 1.0, None, 1.0)
    return X, info
