"""Posterior sampling — same call surface as the reference's ``src/sample.py``.

``sample`` (``:148``), ``inv_matsqrt_vp`` (``:55``), ``inv_matsqrt_dense`` (``:16``), ``sample_dense``
(``:159``), ``sample_both`` (``:168``), plus ``sample_lanczos`` (D-space Lanczos on GGN + alpha I, the
variant BASELINE.json's north star names).

A = alpha I + beta W W^T (W built at N/M = 1, beta = N/M applied here: ``:64,109-111``):
  A^(-1/2) v = W (W^T W)^+ f(alpha I_d + beta W^T W) W^T v + alpha^(-1/2) (v - W (W^T W)^+ W^T v),
f(x) = x^(-1/2) evaluated by a min(2M, d)-step Lanczos in the small d = M K space (``:113-128``).
W is linear, so both terms share ONE W^T sweep and ONE W sweep per sample block (the reference does
two of each per sample, sequentially: ``:78-85,130-139,155``).

Decisions on the reference's defects (documented in DESIGN.md):
  * ``clip_min``: the reference evaluates f(max(lambda, 1)) through its monkey-patched eigh
    (``src/matfree_monkeypatch.py:19``); default here is the mathematics (``None``); pass
    ``clip_min=REFERENCE_CLIP_MIN`` for the reference's behaviour.
  * ``method``: the reference evaluates f(alpha I_d + beta W^T W) u by a min(2M, d)-step Lanczos
    (``"lanczos"``, kept, on the HIP Krylov kernels).  At the CIFAR config that matrix has condition
    number ~1e9 and 2M = 100 steps do not resolve x^(-1/2) on it, so the default is the exact
    evaluation through one eigendecomposition of the d x d matrix (``"eigh"``) — cheaper as well.
  * (W^T W) is singular for the classifier (rank K-1 per example, §4.1-5) and whenever d > D; the
    reference calls ``solve`` on it.  Here the Moore-Penrose pseudo-inverse is used (identical when
    W^T W is invertible).
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import krylov
from .ggn import (FACTOR_BYTES_LIMIT, BlockOperator, build_WTW, compute_ggn_vp, compute_W_vps, gram_from_factor,
                  materialize_factor)
from .utils import flatten_nn_params

REFERENCE_CLIP_MIN = 1.0


def _seed(key) -> int:
    if key is None:
        return 0
    if torch.is_tensor(key):
        return int(key.reshape(-1)[0].item())
    return int(key)


# Eigenvalues of the Gram W^T W at or below rtol * max are treated as exact zeros (the classifier's Gram has rank
# M (K - 1)).  A dropped direction is sampled with the PRIOR variance 1/alpha instead of
# 1/(alpha + beta lambda), so the threshold has to sit at the noise floor of the Gram, not above it: 1e-12 for the
# float64-accumulated Gram of the materialised float32 factor (its null eigenvalues come out at ~eps_f32^2 = 4e-15 of
# the top one), 1e-6 for a Gram assembled from float32 engine outputs (matrix-free route; noise ~1e-6 of the top one).
# (With 1e-6 everywhere the XOR config, alpha = 9e-4, drew 1 % too much energy: beta lambda of the dropped directions
# was up to ten times alpha.)
GRAM_RTOL = 1e-12
GRAM_RTOL_F32 = 1e-6


def _psd_and_pinv(G: torch.Tensor, rtol: float = GRAM_RTOL, return_eig: bool = False):
    """Project the symmetric Gram matrix onto the PSD cone and return (G_psd, G^+) in float64.
    Eigenvalues <= rtol * max are treated as zero (the classifier's W^T W has rank M (K-1))."""
    ev, U = torch.linalg.eigh(G.double())
    keep = ev > rtol * ev.max().clamp_min(1e-300)
    evp = torch.where(keep, ev, torch.zeros_like(ev))
    inv = torch.where(keep, 1.0 / ev.clamp_min(1e-300), torch.zeros_like(ev))
    out = ((U * evp) @ U.T, (U * inv) @ U.T)
    return out + (evp, U) if return_eig else out


def _pinv_sym(G: torch.Tensor, rtol: float = GRAM_RTOL) -> torch.Tensor:
    return _psd_and_pinv(G, rtol)[1]


def orthonormal_factor(Wm: torch.Tensor, ev: torch.Tensor, Ug: torch.Tensor, keep: torch.Tensor) -> torch.Tensor:
    """Qm (r, D): rows q_k = lambda_k^(-1/2) sum_i Ug[i, k] Wm[i] for the kept eigenpairs of the Gram Wm Wm^T = Ug diag(ev)
    Ug^T — an orthonormal basis of range(W) in which W W^T = sum_k lambda_k q_k q_k^T.  The combination is formed in
    float64 (column slabs of the factor, one float64 GEMM each) and rounded once: a float32 GEMM would leave the stiff
    directions accurate to ~1e-6 only, and the sampler needs them to ~1e-7 (see ``_SamplerParts``)."""
    C = (Ug[:, keep] * torch.rsqrt(ev[keep])).T.contiguous()              # (r, d) float64
    r, D = C.shape[0], Wm.shape[1]
    Qm = torch.empty(r, D, device=Wm.device, dtype=torch.float32)
    slab = 1 << 16
    for c in range(0, D, slab):
        Qm[:, c:c + slab] = (C @ Wm[:, c:c + slab].double()).float()
    return Qm


class _SamplerParts:
    """Everything ``inv_matsqrt_vp`` precomputes once per (state, Z, alpha).

    ``method="eigh"`` with a materialised factor applies A^(-1/2) in the eigenbasis of the Gram: with
    W W^T = sum_k lambda_k q_k q_k^T (q_k orthonormal, the rows of ``Qm``) the reference's formula
    (``src/sample.py:78-85,130-143``) collapses to

        A^(-1/2) v = alpha^(-1/2) v + sum_k q_k (f(alpha + beta lambda_k) - alpha^(-1/2)) <q_k, v> .

    The literal form W (W^T W)^+ [f(A_d) - alpha^(-1/2)] W^T v is numerically hopeless in float32 at the CIFAR config
    (cond(A) = 3e9): the d x d matrix in the middle spans six decades, so its float32 image no longer contains the
    stiff directions at all, and the stiff component of a draw — 1.8e-5 of the alpha^(-1/2) v it cancels against —
    came out with O(1) relative error (measured through the matrix-free operator: whitening error 2.4e-3 of ||v||^2;
    4e-5 in this form).  Here every quantity is O(|v|): the coefficients <q_k, v> come from one float32 GEMM, are
    scaled in float64, and the second GEMM adds the correction to alpha^(-1/2) v in its epilogue."""

    def __init__(self, state, Z, D, alpha, model_type, full_set_size, clip_min, method):
        self.Wfun, self.WTfun = compute_W_vps(state, Z, model_type, full_set_size=None)   # :64
        eng = self.Wfun.engine
        self.eng, self.alpha = eng, float(alpha)
        M = Z.shape[0]
        N = full_set_size or M
        self.beta = N / M
        self.inner = self.WTfun.out_shape
        self.d = math.prod(self.inner)
        # In the inducing-point regime the factor Wm (d, D) fits in HBM: materialise it once (d backward rows),
        # take the Gram from it and apply W^T / W as plain GEMMs.  Otherwise stay matrix-free throughout.
        self.Wm = self.Qm = None
        if self.d * eng.D * 4 <= FACTOR_BYTES_LIMIT:
            c = math.sqrt(1.0) * (math.exp(-0.5 * float(state.params["logvar"]["logvar"])) if model_type == "regressor" else 1.0)
            self.Wm = materialize_factor(eng, c)
            G64 = gram_from_factor(self.Wm)
            G64 = torch.triu(G64) + torch.triu(G64, 1).T                                                 # :227
        else:
            G64 = build_WTW(self.Wfun, self.WTfun, self.inner, self.d, dtype=torch.float64, block=2)    # :77
        G_psd, G_pinv, evp, Ug = _psd_and_pinv(G64, GRAM_RTOL if self.Wm is not None else GRAM_RTOL_F32, return_eig=True)
        self.WTW = G_psd.float().contiguous()
        self.G_pinv = G_pinv.float().contiguous()
        A64 = self.alpha * torch.eye(self.d, device=eng.device, dtype=torch.float64) + self.beta * G_psd
        self.A_d = A64.float().contiguous()
        self.depth = min(2 * M, self.d)                                                     # :114
        self.clip_min, self.method = clip_min, method
        f = lambda x: 1.0 / torch.sqrt(x)
        if method == "lanczos":
            self.funm = krylov.funm_lanczos_dense(krylov.dense_funm_sym_eigh(f, clip_min, floor=self.alpha), self.depth)   # :113-115
            self.A_d64, self.G_pinv64 = A64.contiguous(), G_pinv.contiguous()
        elif method == "eigh":
            # alpha I + beta G_psd shares G_psd's eigenvectors: f(A) from the decomposition already at hand
            keep = evp > 0
            lamA = torch.clamp(self.alpha + self.beta * evp, min=self.alpha)
            if clip_min is not None:
                lamA = torch.clamp(lamA, min=clip_min)
                g = f(lamA) - 1.0 / math.sqrt(self.alpha)
            else:
                # f(alpha + beta lambda) - alpha^(-1/2) in its cancellation-free form
                sa, sl = math.sqrt(self.alpha), torch.sqrt(lamA)
                g = -self.beta * evp / (sa * sl * (sa + sl))
            # the same operator in W^T-coordinates, for the matrix-free fallback:  X = U Mc,  out = W X + a v
            h = torch.where(keep, g / evp.clamp_min(1e-300), torch.zeros_like(g))
            self.Mc64 = ((Ug * h) @ Ug.T).contiguous()
            order = torch.argsort(g[keep])                  # most negative first = stiffest direction first
            kept = torch.nonzero(keep).flatten()[order]     # index form: rows of Qm sorted by stiffness
            self.g = g[kept].contiguous()                                                   # (r,) float64
            # <q_k, v> is needed to alpha^(1/2) (alpha + beta lambda_k)^(-1/2) * tol RELATIVE accuracy (the draw's stiff
            # component is that fraction of the alpha^(-1/2) <q_k, v> it cancels against): a float32-accumulated GEMM
            # over D = 1e6 delivers ~4e-6, enough for directions with sqrt((alpha + beta lambda) / alpha) < 250 at
            # tol = 1e-3; the stiffer ones (at most 96, first in the sorted order) get float64-accumulated coefficients
            stiff = torch.sqrt(lamA[kept] / self.alpha) > 250.0
            self.n_stiff = int(min(96, int(stiff.sum().item())))
            if self.Wm is not None:
                self.Qm = orthonormal_factor(self.Wm, evp, Ug, kept)
        else:
            raise ValueError("method must be 'lanczos' or 'eigh'")

    def f_small(self, U: torch.Tensor) -> torch.Tensor:
        """f(alpha I + beta W^T W) applied to the rows of U (S, d) float64 — the reference's small-space Lanczos (:117-128)."""
        return self.funm(self.A_d64, U)

    def _correction(self, V: torch.Tensor):
        """(T, B) with  A^(-1/2) V = alpha^(-1/2) V + T B  (T small and float32, B the (rows, D) factor it multiplies)."""
        S = V.shape[0]
        a = 1.0 / math.sqrt(self.alpha)
        if self.method == "eigh" and self.Qm is not None:
            # <q_k, v>: one pass over the factor — the split-K MFMA kernel from 32 draws up (2.8 ms against hipBLASLt's
            # 6.4 ms at 256 x 450 x 1.08 M; below that its 128-row tiles run mostly empty and the library wins)
            C = (krylov.gemm_nt(V, self.Qm) if V.shape[0] >= 32 else V @ self.Qm.T).double()
            if self.n_stiff:
                C[:, :self.n_stiff] = krylov.dot_nt(V, self.Qm[:self.n_stiff])             # float64-accumulated
            return (C * self.g).float().contiguous(), self.Qm
        if self.Wm is not None:
            U = V @ self.Wm.T                                                              # W^T v as a GEMM
        else:
            U = self.WTfun.rows(V).reshape(S, self.d)                                      # W^T v, matrix-free
        if self.method == "eigh":
            X = (U.double() @ self.Mc64).float().contiguous()                              # :78-84 and :130-138 fused
        else:
            Ud = U.double()                                 # the d x d algebra spans nine decades: float64
            x1 = self.f_small(Ud) @ self.G_pinv64                                          # :130-138
            x2 = Ud @ self.G_pinv64                                                        # :78-84
            X = (x1 - a * x2).float().contiguous()
        return X, self.Wm

    def _degenerate(self) -> bool:
        """no direction of positive curvature kept (a zero Jacobian): A = alpha I"""
        return self.method == "eigh" and self.Qm is not None and self.Qm.shape[0] == 0

    def apply(self, V: torch.Tensor) -> torch.Tensor:
        """rows of V (S, D) -> A^(-1/2) V"""
        if self._degenerate():
            return V / math.sqrt(self.alpha)
        T, B = self._correction(V)
        a = 1.0 / math.sqrt(self.alpha)
        if B is not None:
            return _nn_axpy(T, B, V, a)                                                    # W x + alpha^(-1/2) v in one pass
        out = self.Wfun.rows(T.reshape((V.shape[0],) + self.inner))                        # one W sweep, matrix-free
        return krylov.axpby(out, V.contiguous(), None, a, None, 1.0)                       # + alpha^(-1/2) v

    def apply_(self, V: torch.Tensor) -> torch.Tensor:
        """In-place :meth:`apply` for the materialised-factor case: the ``+ alpha^(-1/2) v`` term rides in the second
        GEMM's epilogue (``beta * C``), so a block of draws costs two GEMM passes over the factor and nothing else."""
        if self._degenerate():
            return V.mul_(1.0 / math.sqrt(self.alpha))
        T, B = self._correction(V)
        if B is None:
            V.copy_(self.apply(V))
            return V
        return _nn_axpy(T, B, V, 1.0 / math.sqrt(self.alpha), out=V)


def _nn_axpy(T, B, V, beta, out=None):
    """beta V + T B, the second pass of a block of draws.  Measured on MI355X at (256 x 450)(450 x 1.08 M)
    (``scripts/sampler_gemm_bench.py``, round 3): the build's own NN kernel ``lip_gemm_nn_axpy`` 2.91 ms out of place and
    3.00 ms in place (86 TFLOP/s); hipBLASLt through ``torch.addmm`` 2.82 ms out of place and 2.13 ms in place
    (117 TFLOP/s: the addend is the output, one pass over the block fewer).  The library stays on this path; the own
    kernel is selected with ``LIP_OWN_GEMM=1`` (A/B) and is what a build without hipBLASLt would run."""
    if _OWN_GEMM:
        return krylov.gemm_nn_axpy(T, B, V, beta, out=out)
    return torch.addmm(V, T, B, beta=beta, alpha=1.0) if out is None else torch.addmm(V, T, B, beta=beta, alpha=1.0, out=out)


import os as _os
_OWN_GEMM = bool(_os.environ.get("LIP_OWN_GEMM"))
_PARTS_CACHE = {}
_PARTS_CACHE_MAX = 2


def _drop_parts_of(engine_key):
    """An evicted engine takes its sampler parts (the multi-GB factor, the Gram, f(A)) with it."""
    for k in [k for k in _PARTS_CACHE if k[0] == engine_key]:
        _PARTS_CACHE.pop(k)


def _cached_parts(state, Z, D, alpha, model_type, full_set_size, clip_min, method):
    """One factor / Gram / f(A) build per (engine binding, alpha, N, clip, method): evaluation loops call
    ``predict_lla_scalable`` once per test batch with the same (state, Z) (``scale_experiments/evaluate.py:103``),
    and the reference rebuilds everything each time (``src/lla.py:137``).  Keyed on the engine's cache key and LRU
    like the engine cache; parts die with their engine."""
    from . import ggn as _g
    if _drop_parts_of not in _g._EVICTION_HOOKS:
        _g._EVICTION_HOOKS.append(_drop_parts_of)
    eng = _g.get_engine(state, Z, model_type)
    key = (eng.cache_key, float(alpha), full_set_size, clip_min, method)
    parts = _PARTS_CACHE.pop(key, None)
    if parts is None or parts.eng is not eng:
        parts = _SamplerParts(state, Z, D, alpha, model_type, full_set_size, clip_min, method)
        while len(_PARTS_CACHE) >= _PARTS_CACHE_MAX:
            _PARTS_CACHE.pop(next(iter(_PARTS_CACHE)))
    _PARTS_CACHE[key] = parts
    return parts


def _compat(reference_compat: bool, clip_min, method):
    """``reference_compat=True`` = the reference's own numerics in one switch: the monkey-patched eigenvalue clip
    f(max(lambda, 1)) (``src/matfree_monkeypatch.py:19``) and the min(2M, d)-step small-space Lanczos
    (``src/sample.py:113-115``) instead of the exact eigen-evaluation; ``solve(W^T W, .)`` (``:81-84``) is the
    truncated pseudo-inverse here, identical wherever the Gram is invertible (where it is not — every classifier — the
    reference's solve is ill-posed).  PARITY UNPINNED: no reference test that exercises the clip can pass
    (SURVEY §4.1-7); checked against ``oracle/sample.py`` run with the same flags."""
    return (REFERENCE_CLIP_MIN, "lanczos") if reference_compat else (clip_min, method)


def inv_matsqrt_vp(state, Z, D, alpha, model_type, full_set_size=None, key=None, num_proj_steps=1,
                   clip_min: Optional[float] = None, method: str = "eigh", reference_compat: bool = False):
    """``src/sample.py:55-145``.  Returns a block operator v -> A^(-1/2) v on (D,) or (S, D).
    (``key`` / ``num_proj_steps`` select the reference's alternating-projection branch, which it
    disables itself — ``:150`` forces ``key=None`` because the branch returns NaN)
    ``reference_compat``: see :func:`_compat`."""
    clip_min, method = _compat(reference_compat, clip_min, method)
    parts = _cached_parts(state, Z, D, alpha, model_type, full_set_size, clip_min, method)
    eng = parts.eng
    op = BlockOperator(lambda V: parts.apply(V.to(device=eng.device, dtype=torch.float32).contiguous()),
                       (eng.D,), (eng.D,), eng, "inv_matsqrt_vp")
    op.parts = parts
    return op


def sample(state, Z, D, alpha, key, model_type, num_samples=1, full_set_size=None, num_proj_steps=10,
           clip_min: Optional[float] = None, method: str = "eigh", block: int = 256, reference_compat: bool = False):
    """``src/sample.py:148-156``: ``num_samples`` zero-mean draws A^(-1/2) eps, eps ~ N(0, I) -> (S, D).
    (theta_MAP is *not* added, as in the reference: ``:153-154``.)  eps comes from the in-kernel
    Philox generator seeded by ``key``; bit parity with JAX's threefry is not attempted."""
    fun = inv_matsqrt_vp(state, Z, D, alpha, model_type, full_set_size=full_set_size, key=None,
                         num_proj_steps=num_proj_steps, clip_min=clip_min, method=method, reference_compat=reference_compat)
    eng = fun.engine
    out = torch.empty(num_samples, eng.D, device=eng.device, dtype=torch.float32)
    for s in range(0, num_samples, block):
        e = min(num_samples, s + block)
        krylov.fill_normal(e - s, eng.D, _seed(key) * 1000003 + s, eng.device, out=out[s:e])
        fun.parts.apply_(out[s:e])
    return out


def range_deflation(state, Z, D, alpha, model_type, full_set_size=None) -> "krylov.RangeDeflation":
    """The invariant subspace range(W) of A = alpha I + beta W W^T as the D-space Krylov routes use it: the sampler's
    orthonormalised factor (rows q_k, A q_k = (alpha + beta lambda_k) q_k).  Needs the factor to fit in HBM."""
    parts = _cached_parts(state, Z, D, alpha, model_type, full_set_size, None, "eigh")
    if parts.Qm is None:
        raise ValueError("range deflation needs the materialised factor (d * D * 4 bytes <= FACTOR_BYTES_LIMIT)")
    lamA = (1.0 / (parts.g + 1.0 / math.sqrt(parts.alpha))) ** 2            # alpha + beta lambda_k, in Qm's row order
    return krylov.RangeDeflation(parts.Qm, lamA)


def sample_lanczos(state, Z, D, alpha, key, model_type, num_samples=1, full_set_size=None, num_matvecs=36,
                   clip_min: Optional[float] = None, deflate: bool = False, eps: Optional[torch.Tensor] = None):
    """D-space variant: (GGN + alpha I)^(-1/2) eps by ``num_matvecs``-step Lanczos with full
    re-orthogonalisation on the matrix-free GGN-vector product (the Krylov loop of BASELINE.json's north
    star; the reference only runs its Lanczos in the small d-space).

    ``deflate=True``: range(W) is taken out of the recurrence and handled exactly (``krylov.RangeDeflation``) — the
    float32 product's rounding noise lives in that subspace and is as large as alpha at the CIFAR config's alpha =
    0.005, which is what made the plain recurrence lose accuracy with MORE steps there."""
    vp = compute_ggn_vp(state, Z, model_type, full_set_size=full_set_size)
    eng = vp.engine
    M = Z.shape[0]
    N = full_set_size or M
    scale = N / M * (math.exp(-float(state.params["logvar"]["logvar"])) if model_type == "regressor" else 1.0)
    matvec = lambda V: eng.ggn_vp(V, scale, float(alpha))
    f = lambda x: 1.0 / torch.sqrt(x)
    funm = krylov.funm_lanczos_sym(krylov.dense_funm_sym_eigh(f, clip_min, floor=float(alpha)), num_matvecs)
    Eps = krylov.fill_normal(num_samples, eng.D, _seed(key) * 1000003, eng.device) if eps is None else eps
    if not deflate:
        return funm(matvec, Eps)
    defl = range_deflation(state, Z, D, alpha, model_type, full_set_size)
    C = defl.coeffs(Eps)
    Xp = defl.project_out(funm(defl.wrap(matvec), defl.project_out(Eps, C)))
    fr = (lambda lam: f(torch.clamp(lam, min=clip_min))) if clip_min is not None else f
    return krylov.axpby(defl.range_part(C, fr), Xp, None, 1.0, None, 1.0)


def inv_matsqrt_dense(state, Z, D, alpha, model_type, full_set_size=None):
    """``src/sample.py:16-52`` (debug twin): materialise W (D, d) and use eigh."""
    Wfun, WTfun = compute_W_vps(state, Z, model_type, full_set_size=None)
    eng = Wfun.engine
    Dn = eng.D
    M = Z.shape[0]
    N = full_set_size or M
    beta = N / M
    d = math.prod(WTfun.out_shape)
    E = torch.eye(d, device=eng.device, dtype=torch.float32).reshape((d,) + WTfun.out_shape)
    W = Wfun.rows(E).T.double()                                   # (D, d)
    composite = W.T @ W
    inv_composite = _pinv_sym(composite)
    I_D = torch.eye(Dn, device=eng.device, dtype=torch.float64)
    nullproj = I_D - W @ inv_composite @ W.T
    evals, evecs = torch.linalg.eigh(alpha * torch.eye(d, device=eng.device, dtype=torch.float64) + beta * composite)
    inv_sqrt_term = (evecs * (1.0 / torch.sqrt(torch.clamp(evals, min=1e-300)))) @ evecs.T
    return (nullproj / math.sqrt(alpha) + W @ inv_composite @ inv_sqrt_term @ W.T).float()


def sample_dense(state, Z, D, alpha, key, model_type, num_samples=1, full_set_size=None):
    """``src/sample.py:159-165`` (adds theta_MAP, unlike ``sample``)."""
    A = inv_matsqrt_dense(state, Z, D, alpha, model_type, full_set_size=full_set_size)
    flat_params, _ = flatten_nn_params(state.params)
    Eps = krylov.fill_normal(num_samples, A.shape[0], _seed(key) * 1000003, A.device)
    return Eps @ A.T + flat_params.to(A.device, torch.float32)


def sample_both(state, Z, D, alpha, key, model_type, num_samples=1, full_set_size=None,
                clip_min: Optional[float] = None, method: str = "eigh", reference_compat: bool = False):
    """``src/sample.py:168-178``: the matrix-free and the dense sampler on the same noise."""
    fun = inv_matsqrt_vp(state, Z, D, alpha, model_type, full_set_size=full_set_size, clip_min=clip_min, method=method,
                         reference_compat=reference_compat)
    Eps = krylov.fill_normal(num_samples, fun.engine.D, _seed(key) * 1000003, fun.engine.device)
    samples = fun.rows(Eps)
    A = inv_matsqrt_dense(state, Z, D, alpha, model_type, full_set_size=full_set_size)
    return samples, Eps @ A.T
