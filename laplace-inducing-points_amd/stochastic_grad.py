"""Matrix-free gradient of the reference's STOCHASTIC inducing-point objective
(``jax.value_and_grad(alternative_objective_scalable)``, ``/root/reference/src/train_inducing.py:87-173,195-196``).

The objective is  F(Z) = hutchpp_v2(C; probes) + mean_p SLQ_k(p),  C = S S_z^-1  with

    S      = alpha I + (N/K_b) GGN(X)                         data precision (does not depend on Z)
    S_z^-1 = alpha^-1 I - alpha^-2 W M^-1 W^T =: B            Woodbury, M = beta^-1 I + alpha^-1 W^T W   (``:127-132``)
    SLQ_k  = ||p||^2 e1^T log(B_k^T B_k) e1,  B_k the Golub-Kahan bidiagonal of  v -> [sqrt(alpha) v ; s_b W^T v]  (``:156-169``)

and it depends on Z only through the factor W = W(Z) (D x d, d = M K; column (j, k) = c J(z_j)^T L(z_j) e_k).  Reverse
mode through the two estimators therefore ends in a cotangent of W that is a sum of rank-one terms,

    dF = sum_t u_t^T dW x_t ,        u_t in R^D,  x_t in R^d ,

one or two per operator application of the forward pass, and the gradient is that of the PAIRING
sum_t sum_j < J(z_j) u_t , c L(z_j) x_tj >  with (u_t, x_t) frozen — the second-order pass of ``second_order.py`` with the
directions shared by the examples.  No factor of the data batch (K rows of D floats per image) is formed: X enters
only through products with S, twice as many as the value alone needs.

Hutch++ (``src/stochtrace.py:118-135``).  With Q an orthonormal basis of range(Y), Y = C S_1^T, the estimator is a
function of C and of the projector Pi = Q Q^T only,  T = tr(C Pi) + (1/s2) tr(G (I - Pi) C (I - Pi) G^T), so its
gradient does not depend on which QR produced Q (the reference's Householder QR, ``:128``, or the CholeskyQR2 of
``krylov.gram_orthonormalize``).  dPi = (I - Pi) dY Y^+ + transpose gives, in row form (rows = vectors),

    Ybar = Cm^T [ E_perp - (1/s2) ( (H Q^T)^T G_perp + (G Q^T)^T H_perp ) ],   E = (C + C^T) Q,  H = (C + C^T) G_perp,

(Q = Cm Y; _perp = rows projected off range(Q)) and the cotangent of C is  sum_i ybar_i s_i^T + sum_i q_i q_i^T +
(1/s2) sum_r g_r g_r^T  (g_r the projected probes).  For C = S B, a term a b^T of Cbar contributes

    -alpha^-1 [ (B S a) (M^-1 W^T b)^T + (B b) (M^-1 W^T S a)^T ]

to the cotangent of W (the dependence of M^-1 on W^T W included) — every vector on the left already exists in the
forward / adjoint pass.

SLQ.  Reverse mode runs over the Golub-Kahan recurrence exactly as it ran, re-orthogonalisation passes included (their
projection coefficients are kept by the forward pass): one application of A and one of A^T per step, two rank-one
terms per step.  Transposing only the plain three-term recurrence would give the same gradient in exact arithmetic
(the passes are then the identity) but an unstable backward recurrence: measured and rejected, see the code.

The D-sized linear algebra is behind a small vector backend: :class:`HipVec` (``lip_dot_nt_f64``, ``lip_rows_combine``,
``lip_bdot``, ``lip_axpby`` — the product path) and :class:`TorchVec` (plain torch, any dtype / device: used by the CPU
tests of the host logic against reverse mode through the oracle).
"""
from __future__ import annotations

import math
from typing import Callable, List, Tuple

import torch


# ----------------------------------------------------------------------------------------------------------------
# vector backends
# ----------------------------------------------------------------------------------------------------------------
class TorchVec:
    """Plain torch algebra on (rows, N) blocks (tests of the host logic; float64 on the CPU)."""

    def gram(self, A, B):                                   # A B^T, float64
        return A.double() @ B.double().T

    def combine(self, C, Y, Z=None, zscale=0.0):            # C Y + zscale Z
        out = C.to(Y.dtype) @ Y
        return out if Z is None else out + zscale * Z

    def bdot(self, X, Y):                                   # row-wise <x, y>, float64
        return (X.double() * Y.double()).sum(1)

    def lin(self, terms):                                   # sum_i coef_i[:, None] * block_i ; coef (P,) or scalar
        out = None
        for c, X in terms:
            c = c if not torch.is_tensor(c) else c.to(X.dtype)[:, None]
            out = c * X if out is None else out + c * X
        return out

    def orth(self, Y):                                      # Q (r, N) orthonormal rows, Cm with Q = Cm Y
        Qc, R = torch.linalg.qr(Y.T, mode="reduced")        # Y^T = Qc R  ->  Q = R^-T Y
        Cm = torch.linalg.solve_triangular(R.T, torch.eye(R.shape[0], dtype=Y.dtype, device=Y.device), upper=False)
        return Qc.T.contiguous(), Cm.double()

    def bidiag(self, matvec, vecmat, V0, k, n_out):
        """Golub-Kahan with full re-orthogonalisation (classical Gram-Schmidt twice against ALL previous vectors), P
        recurrences at once: (alphas, betas, V (P,k,N), U (P,k,n_out), coeffs) with coeffs = (cu1, cu2, cv1, cv2), each
        (P, k, k): row j holds the projection coefficients the two passes of step j took out (the adjoint needs them)."""
        P, N = V0.shape
        kw = dict(dtype=V0.dtype, device=V0.device)
        V, U = torch.zeros(P, k, N, **kw), torch.zeros(P, k, n_out, **kw)
        al, be = torch.zeros(P, k, **kw), torch.zeros(P, max(k - 1, 0), **kw)
        cu1, cu2, cv1, cv2 = (torch.zeros(P, k, k, **kw) for _ in range(4))
        v = V0 / V0.norm(dim=1, keepdim=True)
        for j in range(k):
            V[:, j] = v
            u = matvec(v)
            for c in (cu1, cu2):
                c[:, j, :j] = torch.einsum("pkn,pn->pk", U[:, :j], u)
                u = u - torch.einsum("pk,pkn->pn", c[:, j, :j], U[:, :j])
            al[:, j] = u.norm(dim=1)
            u = u / al[:, j, None]
            U[:, j] = u
            if j + 1 < k:
                w = vecmat(u)
                for c in (cv1, cv2):
                    c[:, j, :j + 1] = torch.einsum("pkn,pn->pk", V[:, :j + 1], w)
                    w = w - torch.einsum("pk,pkn->pn", c[:, j, :j + 1], V[:, :j + 1])
                be[:, j] = w.norm(dim=1)
                v = w / be[:, j, None]
        return al, be, V, U, (cu1, cu2, cv1, cv2)

    # batched Gram-Schmidt pieces of the adjoint recurrence, Q (P, j, N) a block of basis vectors
    def proj_coeffs(self, Q, x):                            # (P, j) = Q x
        return torch.einsum("pkn,pn->pk", Q, x)

    def sub_combination(self, x, c, Q):                     # x - c Q
        return x - torch.einsum("pk,pkn->pn", c.to(x.dtype), Q)

    def add_combination(self, x, c, Q):                     # x + c Q
        return x + torch.einsum("pk,pkn->pn", c.to(x.dtype), Q)

    def rank1_update_(self, Bar, a, x):                     # Bar (P, j, N) -= a (P, j) (x) x (P, N)
        Bar.baddbmm_(a.to(x.dtype)[:, :, None], x[:, None, :], alpha=-1.0)


class HipVec(TorchVec):
    """The product backend: the D-sized operations are kernels of ``csrc/lip_krylov.hip``, the projections of the SLQ
    adjoint included; only its rank-one updates of a (P, j, N) block of basis cotangents (an HBM-bound batched GER) run
    as a torch batched product on the device."""

    def __init__(self):
        from . import krylov
        self.k = krylov

    def gram(self, A, B):
        return self.k.dot_nt(A.contiguous(), B.contiguous())

    def combine(self, C, Y, Z=None, zscale=0.0):
        return self.k.rows_combine(C, Y.contiguous(), None if Z is None else Z.contiguous(), zscale)

    def bdot(self, X, Y):
        return self.k.bdot(X.contiguous(), Y.contiguous()).double()

    def lin(self, terms):
        (c0, X0), rest = terms[0], terms[1:]
        P = X0.shape[0]
        dev = X0.device

        def coef(c):
            return c.to(device=dev, dtype=torch.float32).contiguous() if torch.is_tensor(c) else torch.full((P,), float(c), device=dev)

        out = X0.contiguous().clone()
        zero = torch.zeros(P, device=dev)
        self.k.axpby(out, out, zero, 1.0, coef(c0), 1.0)                     # out = c0 * X0
        one = torch.ones(P, device=dev)
        for c, X in rest:
            self.k.axpby(out, X.contiguous(), coef(c), 1.0, one, 1.0)        # out += c * X
        return out

    def orth(self, Y):
        return self.k.gram_orthonormalize(Y.contiguous(), return_transform=True)

    def bidiag(self, matvec, vecmat, V0, k, n_out):
        return self.k.bidiag(matvec, vecmat, V0.contiguous(), k, n_out, return_bases=True)     # (..., V, U, coefficients)

    # The batched projections as library einsums fall to a generic batched-GEMM kernel once the inner dimension is the
    # parameter count (hipBLASLt refuses k = 25.6 M: 125 ms per call, 4 s of the ResNet-50 step); per recurrence p they
    # are exactly dot_nt (j x N against 1 x N, float64 accumulation) and rows_combine (one row out of j), which take
    # row-strided views of the basis block as they are.
    def proj_coeffs(self, Q, x):
        return torch.stack([self.k.dot_nt(Q[p], x[p:p + 1])[:, 0] for p in range(Q.shape[0])]).to(x.dtype)

    def sub_combination(self, x, c, Q):
        return torch.cat([self.k.rows_combine(-c[p:p + 1].double(), Q[p], x[p:p + 1], 1.0) for p in range(Q.shape[0])])

    def add_combination(self, x, c, Q):
        return torch.cat([self.k.rows_combine(c[p:p + 1].double(), Q[p], x[p:p + 1], 1.0) for p in range(Q.shape[0])])


# ----------------------------------------------------------------------------------------------------------------
# Hutch++ on C = S B: value and cotangent of W
# ----------------------------------------------------------------------------------------------------------------
def _hutchpp_value_and_terms(S_rows, WT_rows, W_rows, Minv, alpha, probes, s1, s2, vec):
    a_inv = 1.0 / alpha

    def B(V):                                   # S_z^-1 on rows (Woodbury, :127-132); also x = M^-1 W^T v
        x = WT_rows(V).double() @ Minv
        return vec.lin([(a_inv, V), (-a_inv * a_inv, W_rows(x.to(V.dtype)))]), x

    Sp, G = probes[:s1], probes[s1:s1 + s2]
    # ---- forward (src/stochtrace.py:118-135)
    BS, xS = B(Sp)
    Y = S_rows(BS)
    Q, Cm = vec.orth(Y)                         # Q = Cm Y, orthonormal rows
    r = Q.shape[0]
    BQ, xQ = B(Q)
    CQ = S_rows(BQ)
    low_rank = vec.bdot(Q, CQ).sum()
    GQ = vec.gram(G, Q)                         # (s2, r)
    Gp = vec.combine(-GQ, Q, G, 1.0)            # G - (G Q^T) Q
    BG, xG = B(Gp)
    CG = S_rows(BG)
    resid = vec.bdot(Gp, CG).sum() / s2
    value = float(low_rank + resid)
    # ---- adjoint
    CtQ, yQ = B(S_rows(Q))                      # C^T q = B S q ;  y = M^-1 W^T S q
    CtG, yG = B(S_rows(Gp))
    E = vec.lin([(1.0, CQ), (1.0, CtQ)])
    H = vec.lin([(1.0, CG), (1.0, CtG)])
    Ep = vec.combine(-vec.gram(E, Q), Q, E, 1.0)
    HQ = vec.gram(H, Q)
    Hp = vec.combine(-HQ, Q, H, 1.0)
    coef = torch.cat([HQ.T, GQ.T], dim=1) * (-1.0 / s2)            # (r, 2 s2)
    YbarR = vec.combine(coef, torch.cat([Gp, Hp], dim=0), Ep, 1.0)
    Ybar = vec.combine(Cm.T.contiguous(), YbarR)                    # (s1, D)
    BSY, yY = B(S_rows(Ybar))
    # a b^T in Cbar  ->  -alpha^-1 [ (B S a) (M^-1 W^T b)^T + (B b) (M^-1 W^T S a)^T ]
    m = -a_inv
    terms = [(BSY, m * xS), (CtQ, m * xQ), (CtG, (m / s2) * xG),         # (a, b) = (ybar, s), (q, q), (g/s2, g)
             (BS, m * yY), (BQ, m * yQ), (BG, (m / s2) * yG)]
    return value, terms, dict(rank=r)


# ----------------------------------------------------------------------------------------------------------------
# SLQ on the bidiagonalisation of A v = [sqrt(alpha) v ; s_b W^T v]: value and cotangent of W
# ----------------------------------------------------------------------------------------------------------------
def _slq_small(alphas, betas, len2):
    """mean_p ||p||^2 e1^T log(B^T B) e1 for upper-bidiagonal B (P, k, k), float64, and its gradient w.r.t. the two
    diagonals — in closed form (Daleckii-Krein): with T = B^T B = U diag(ev) U^T and u = U^T e1,

        d (e1^T log(T) e1) = < Phi, dT >,   Phi = U (F o u u^T) U^T,   F_ij = (log ev_i - log ev_j) / (ev_i - ev_j),  F_ii = 1 / ev_i,

    dT = dB^T B + B^T dB  =>  d / dB = 2 B Phi.  Differentiating ``eigh`` itself (autograd) divides by ev_i - ev_j, and
    a Lanczos run that has converged on part of the spectrum delivers clustered Ritz values: NaN at the CIFAR config
    (k = 40).  The divided difference of log is smooth: for close pairs it is evaluated as log1p(x) / x / ev_j, exact
    to rounding for every separation."""
    Bm = torch.diag_embed(alphas)
    if betas.shape[-1] > 0:
        Bm = Bm + torch.diag_embed(betas, 1)
    T = Bm.transpose(-1, -2) @ Bm
    ev, U = torch.linalg.eigh(T)
    ev = ev.clamp_min(1e-300)
    u = U[:, 0, :]                                               # (P, k) = e1^T U
    vals = (u * u * torch.log(ev)).sum(-1)
    value = (len2 * vals).mean()
    x = ev[:, :, None] / ev[:, None, :] - 1.0                    # ev_i / ev_j - 1
    safe = torch.where(x.abs() < 1e-300, torch.ones_like(x), x)
    F = torch.where(x.abs() < 1e-300, torch.ones_like(x), torch.log1p(safe) / safe) / ev[:, None, :]
    F = 0.5 * (F + F.transpose(-1, -2))
    Phi = U @ (F * (u[:, :, None] * u[:, None, :])) @ U.transpose(-1, -2)
    G = 2.0 * (Bm @ Phi) * (len2 / len2.shape[0])[:, None, None]
    gal = torch.diagonal(G, dim1=-2, dim2=-1).clone()
    gbe = torch.diagonal(G, offset=1, dim1=-2, dim2=-1).clone() if betas.shape[-1] > 0 else torch.zeros_like(betas)
    return value, gal, gbe


def _slq_value_and_terms(WT_rows, W_rows, D, d, alpha, s_b, probes, k, vec):
    sa = math.sqrt(alpha)
    dt = probes.dtype

    def A(V):                                   # (P, D) -> (P, D + d)
        return torch.cat([sa * V, s_b * WT_rows(V).to(dt)], dim=1).contiguous()

    def AT(Uu):                                 # (P, D + d) -> (P, D)
        return vec.lin([(sa, Uu[:, :D].contiguous()), (s_b, W_rows(Uu[:, D:].contiguous()))])

    P = probes.shape[0]
    len2 = vec.bdot(probes, probes)
    al, be, Vb, Ub, (cu1, cu2, cv1, cv2) = vec.bidiag(A, AT, probes, k, D + d)
    al64, be64 = al.double(), be.double()
    val, gal, gbe = _slq_small(al64, be64, len2.to(al64.device))
    # ---- reverse mode over the recurrence AS IT RAN: per step  u0 = A v_j, two Gram-Schmidt passes against U_{<j},
    # u_j = u2 / alpha_j;  w0 = A^T u_j, two passes against V_{<=j}, v_{j+1} = w2 / beta_j.  The adjoint of a pass
    # y = x - sum_i (q_i . x) q_i is  xbar = ybar - sum_i (q_i . ybar) q_i  and  qbar_i -= (q_i . x) ybar + (q_i . ybar) x.
    # (Dropping the passes — they are the identity in exact arithmetic, so the FUNCTION is that of the plain three-term
    # recurrence — gives an adjoint recurrence that amplifies rounding by ~10x per step: 0.19 off at k = 20 in float64
    # and overflow at k = 30, NaN at the CIFAR config's k = 40; with the passes transposed it is as stable as reverse
    # mode through the re-orthogonalised algorithm: 4e-4 in float32 at k = 30, same toy — scripts/slq_adjoint_stability.py.)
    Ubar, Vbar = torch.zeros_like(Ub), torch.zeros_like(Vb)
    terms: List[Tuple[torch.Tensor, torch.Tensor]] = []

    def pass_adjoint(Q, Qbar, ybar, c, x_in):
        """one Gram-Schmidt pass y = x_in - c Q, c = Q x_in: returns xbar and updates the basis cotangents Qbar"""
        t = vec.proj_coeffs(Q, ybar)
        vec.rank1_update_(Qbar, c, ybar)
        vec.rank1_update_(Qbar, t, x_in)
        return vec.sub_combination(ybar, t, Q)

    for j in range(k - 1, -1, -1):
        vj, uj = Vb[:, j], Ub[:, j]
        if j + 1 < k:
            vn, bj = Vb[:, j + 1], be64[:, j]
            vb = Vbar[:, j + 1]
            w2bar = vec.lin([(1.0 / bj, vb), (gbe[:, j] - vec.bdot(vn, vb) / bj, vn)])
            Q, Qbar = Vb[:, :j + 1], Vbar[:, :j + 1]
            w1 = vec.add_combination(vec.lin([(bj, vn)]), cv2[:, j, :j + 1], Q)          # input of the second pass
            w0 = vec.add_combination(w1, cv1[:, j, :j + 1], Q)                             # input of the first pass
            w1bar = pass_adjoint(Q, Qbar, w2bar, cv2[:, j, :j + 1], w1)
            w0bar = pass_adjoint(Q, Qbar, w1bar, cv1[:, j, :j + 1], w0)
            Ubar[:, j] += A(w0bar)                                                          # w0 = A^T u_j
            terms.append((w0bar, s_b * uj[:, D:].double()))
        ub = Ubar[:, j]
        u2bar = vec.lin([(1.0 / al64[:, j], ub), (gal[:, j] - vec.bdot(uj, ub) / al64[:, j], uj)])
        if j > 0:
            Q, Qbar = Ub[:, :j], Ubar[:, :j]
            u1 = vec.add_combination(vec.lin([(al64[:, j], uj)]), cu2[:, j, :j], Q)
            u0 = vec.add_combination(u1, cu1[:, j, :j], Q)
            u1bar = pass_adjoint(Q, Qbar, u2bar, cu2[:, j, :j], u1)
            u0bar = pass_adjoint(Q, Qbar, u1bar, cu1[:, j, :j], u0)
        else:
            u0bar = u2bar
        Vbar[:, j] += AT(u0bar.contiguous())                                                # u0 = A v_j
        terms.append((vj.contiguous(), s_b * u0bar[:, D:].double()))
    return float(val), terms


# ----------------------------------------------------------------------------------------------------------------
def stochastic_objective_and_cotangent(S_rows: Callable, WT_rows: Callable, W_rows: Callable, WTW: torch.Tensor, D: int,
                                       alpha: float, beta: float, probes: torch.Tensor, st_samples: int, slq_samples: int,
                                       slq_num_matvecs: int, logdet_beta: bool, vec):
    """Value of ``alternative_objective_scalable`` (``src/train_inducing.py:87-173``) on the given probes and the
    rank-one cotangent of W: ``(value, logdet_term, trace_term, terms)`` with ``dF = sum over (U, X) in terms, rows t:
    U[t]^T dW X[t]``.

    ``S_rows`` (P, D) -> (P, D) applies the data precision; ``WT_rows`` (P, D) -> (P, d) and ``W_rows`` (P, d) -> (P, D)
    the factor of the inducing points (unscaled, ``full_set_size=None`` as at ``:112-114``); ``WTW`` its (d, d) Gram."""
    d = WTW.shape[0]
    I = torch.eye(d, dtype=torch.float64, device=WTW.device)
    Minv = torch.linalg.inv(I / beta + WTW.double() / alpha)
    Minv = 0.5 * (Minv + Minv.T)
    s1, s2 = st_samples - 16, 16                                    # :143
    trace_term, terms, _ = _hutchpp_value_and_terms(S_rows, WT_rows, W_rows, Minv, alpha, probes[:st_samples], s1, s2, vec)
    s_b = math.sqrt(beta) if logdet_beta else 1.0
    logdet_term, t2 = _slq_value_and_terms(WT_rows, W_rows, D, d, alpha, s_b, probes[:slq_samples].contiguous(),
                                           slq_num_matvecs, vec)
    return logdet_term + trace_term, logdet_term, trace_term, terms + t2
