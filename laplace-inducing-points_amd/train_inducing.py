"""Inducing-point KL objective — VALUES of the reference's ``src/train_inducing.py`` objectives ("next" row N1).

``alternative_objective_dense`` (``:176-193``), ``alternative_objective_scalable_exact`` (``:26-84``) and
``alternative_objective_scalable`` (``:87-173``): KL[q(theta|Z) || q(theta|data)] up to constants =
log-det term + trace term, and their gradients w.r.t. Z (``value_and_grad`` at ``:195-196``, ``optimize_step``
``:199-232``): of the exact objective through materialised factors, or — the reference's own quantity — of the
stochastic Hutch++ / SLQ estimate, matrix-free (``stochastic_grad.py``); every D-sized quantity on the HIP engine,
including the last, second-order step (the input derivative of a parameter-JVP: reverse mode over the tangent tape,
``second_order.py``).

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


# ----------------------------------------------------------------------------------------------------------------
# gradients w.r.t. the inducing points (``src/train_inducing.py:195-232``)
# ----------------------------------------------------------------------------------------------------------------
# Both objectives have the form F(Z) = +-[tr(A P_z(Z)^{+-1}) ...] with P_z = alpha I + beta G(Z), G = sum_j W_j W_j^T,
# W_j = c J(z_j)^T L(z_j) (D x K).  Differentiating with the D x D factor Q frozen,
#     dF = beta tr(Q dG) = 2 beta sum_j <Q W_j, dW_j>          =>      grad_{z_j} F = 2 beta grad_z <M_j, W_j(z)>,
# M = Q W (D x d).  Everything D-sized — the factors Wm (d, D) of Z and Wx of the data batch (per-example backward
# sweeps of the HIP engine), their Gram matrices and M (GEMMs) — is exact linear algebra on the device; what is left
# is the derivative of the scalar sum_{j,k} <J(z_j) m_jk, L(z_j) e_k> w.r.t. the inputs z_j: a reverse pass over a
# parameter-JVP, i.e. second order in the network — ``second_order.input_grad_of_pairing``: the tangent tape is run
# with per-example directions and every tangent kept, then walked backwards with the adjoint of each op; all
# convolutions are engine launches (the transposed ones read the HWIO weight tangent in place).
def _c_out(state, model_type):
    return math.exp(-0.5 * float(state.params["logvar"]["logvar"])) if model_type == "regressor" else 1.0


def _gram64(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """A B^T for (a, D) x (b, D) float32 factors with float64 accumulation (``lip_dot_nt_f64``: every product exact; the
    null directions of the Gram meet 1/alpha^2 further on, so float32-accumulated cross-Grams are not an option).
    2.6 x 10^1 TFLOP/s float64 against 7.6 for slabs converted to float64 and multiplied by the library."""
    return krylov.dot_nt(A.contiguous(), B.contiguous())


def _factor_of(state, X, model_type):
    from .ggn import get_engine, materialize_factor
    eng = get_engine(state, X, model_type)
    return eng, materialize_factor(eng, _c_out(state, model_type))


def _input_grad_of_pairing(state, Z, Mrow, model_type):
    """grad_Z of  sum_{j,k} < Mrow[(j,k)], W_(j,k)(z_j) >  =  sum_{j,k} < J(z_j) m_jk, c L(z_j) e_k >: reverse mode over
    the tangent tape on the engine's kernels (``second_order.py``; every convolution / transposed convolution an
    ``LIP_OP_IGEMM`` through ``lip_engine_run_op``)."""
    from .ggn import get_engine
    from .second_order import EngineExecutor, input_grad_of_pairing
    eng = get_engine(state, Z, model_type)
    return input_grad_of_pairing(EngineExecutor(eng), Mrow.reshape(eng.n, eng.K, eng.D), _c_out(state, model_type), model_type)


EXACT_FACTOR_BYTES = 24 << 30      # "auto": the exact route materialises the factor of x_chunk data images (x_chunk K D floats)


def variational_grad_scalable(Z, X, state, alpha, key=None, model_type="classifier", full_set_size=None,
                              x_chunk: Optional[int] = None, method: str = "auto", **kw):
    """``jax.value_and_grad(alternative_objective_scalable)`` (``src/train_inducing.py:196``) -> ``(loss, dLoss/dZ)``.

    ``method="stochastic"`` is the reference's own quantity: value and gradient of the Hutch++ / SLQ ESTIMATE on the
    probes drawn from ``key`` (:func:`variational_grad_stochastic`; matrix-free — the data batch enters through
    products with the data precision only, so it runs where a factor of the batch does not fit: BASELINE configs[4]).
    ``method="exact"``: value and gradient of the exact objective those estimators target, through materialised
    factors (:func:`variational_grad_exact`; no estimator variance, needs x_chunk K D floats per data chunk).
    ``method="auto"`` (default) takes "exact" while the factors of Z and of one data chunk fit ``EXACT_FACTOR_BYTES``
    and "stochastic" beyond that."""
    if method == "auto":
        K, D = _out_dim(state, Z, model_type), _D(state)
        rows = max(Z.shape[0], min(x_chunk or X.shape[0], X.shape[0])) * K
        method = "exact" if rows * D * 4 <= EXACT_FACTOR_BYTES else "stochastic"
    if method == "exact":
        return variational_grad_exact(Z, X, state, alpha, key=key, model_type=model_type, full_set_size=full_set_size,
                                      x_chunk=x_chunk, **{k: v for k, v in kw.items() if k == "_with_constants"})
    if method != "stochastic":
        raise ValueError("method must be 'auto', 'exact' or 'stochastic'")
    skw = {k: v for k, v in kw.items() if k in ("st_samples", "slq_samples", "slq_num_matvecs", "logdet_beta", "probes",
                                                 "example_chunk", "max_directions", "return_terms")}
    return variational_grad_stochastic(Z, X, state, alpha, key=key, model_type=model_type, full_set_size=full_set_size, **skw)


def _out_dim(state, Z, model_type):
    from .ggn import get_engine
    return get_engine(state, Z, model_type).K


def variational_grad_stochastic(Z, X, state, alpha, key=None, model_type="classifier", full_set_size=None, st_samples=256,
                                slq_samples=2, slq_num_matvecs: Optional[int] = None, logdet_beta: bool = True,
                                probes: Optional[torch.Tensor] = None, example_chunk: Optional[int] = None,
                                max_directions: Optional[int] = None, return_terms: bool = False):
    """Value and EXACT gradient of the stochastic objective ``alternative_objective_scalable`` (``src/train_inducing.py:
    87-173``) on fixed probes — what ``jax.value_and_grad`` returns at ``:196``: reverse mode through Hutch++ (the QR
    included) and through the bidiagonalisation SLQ, matrix-free (``stochastic_grad.py`` for the adjoint algebra).

    Cost: 2 (2 s1 + s2) products with the data precision (the value alone needs 2 s1 + s2), (4 s1 + 2 s2 + 4 k slq)
    sweep pairs on the inducing points' engine, one d x d Gram, and one second-order pass over the
    2 (2 s1 + s2) + 2 k slq rank-one directions.  ``example_chunk`` bounds the data images per engine binding
    (``ExampleChunkedGGN``), ``max_directions`` the directions per second-order pass."""
    import time as _time
    from . import stochastic_grad as SG
    from .ggn import ExampleChunkedGGN
    from .second_order import EngineExecutor, input_grad_of_rank_one_terms
    stages, _t = {}, [_time.perf_counter()]

    def _mark(name):                         # stage wall times (synchronised) — only when the caller asks for them
        if return_terms:
            torch.cuda.synchronize()
            now = _time.perf_counter()
            stages[name] = now - _t[0]
            _t[0] = now

    N = full_set_size or Z.shape[0]
    M = Z.shape[0]
    beta = N / M
    if example_chunk is not None and X.shape[0] > example_chunk:
        opS = ExampleChunkedGGN(state, X, model_type, full_set_size=N, example_chunk=example_chunk)
        S_rows = lambda V: opS(V, float(alpha))
    else:
        S_rows = compute_curvature_approx(state, X, alpha=alpha, model_type=model_type, full_set_size=N).rows
    Wz, WzT = compute_W_vps(state, Z, model_type=model_type, full_set_size=None)
    eng = Wz.engine
    D, dev = eng.D, eng.device
    inner = WzT.out_shape
    d_z = math.prod(inner)
    _mark("bind_engines")
    WzTWz = build_WTW(Wz, WzT, inner, d_z, dtype=torch.float64, block=1)
    _mark("gram_WzTWz")
    if probes is None:
        probes = krylov.fill_rademacher(st_samples, D, int(key or 0), dev)        # same probes for both terms, :139-142
    st_samples = probes.shape[0]
    k = slq_num_matvecs if slq_num_matvecs is not None else max(1, int(M * 0.8))   # :148
    WT_rows = lambda V: WzT.rows(V.contiguous()).reshape(V.shape[0], d_z)
    W_rows = lambda Xs: Wz.rows(Xs.to(torch.float32).reshape((Xs.shape[0],) + inner).contiguous())
    if return_terms:                         # per-operator wall time (synchronised) next to the stage times
        def _timed(name, fn):
            def wrapped(*a):
                torch.cuda.synchronize()
                t0 = _time.perf_counter()
                out = fn(*a)
                torch.cuda.synchronize()
                stages["op:" + name] = stages.get("op:" + name, 0.0) + _time.perf_counter() - t0
                stages["n:" + name] = stages.get("n:" + name, 0) + int(a[0].shape[0])
                return out
            return wrapped
        S_rows, WT_rows, W_rows = _timed("S", S_rows), _timed("WzT", WT_rows), _timed("Wz", W_rows)
    value, ld, tr, terms = SG.stochastic_objective_and_cotangent(S_rows, WT_rows, W_rows, WzTWz, D, alpha, beta, probes,
                                                                 st_samples, slq_samples, k, logdet_beta, SG.HipVec())
    _mark("estimators_forward_and_adjoint")
    gZ = input_grad_of_rank_one_terms(EngineExecutor(eng), terms, _c_out(state, model_type), model_type, max_directions)
    gZ = gZ.reshape(Z.shape).to(Z.dtype)
    _mark("second_order_pass")
    if return_terms:
        return value, gZ, dict(logdet_term=ld, trace_term=tr, directions=sum(int(U.shape[0]) for U, _ in terms), stage_seconds=stages)
    return value, gZ


def variational_grad_exact(Z, X, state, alpha, key=None, model_type="classifier", full_set_size=None,
                           x_chunk: Optional[int] = None, _with_constants: bool = False, **_):
    """Value and gradient of the EXACT objective the reference's estimators target.

    The objective is F(Z) = tr(P S_z) + log det P_z  (P = alpha I + gamma G_X the data precision, S_z = P_z^-1), the
    quantity the reference's Hutch++ / SLQ estimators target (``:87-173``; exact twin ``:26-84``).  Returned are its
    EXACT value (through the small Gram matrices, constants dropped as in ``:68,82``) and EXACT gradient, not the
    gradient of a stochastic estimate: in the inducing regime the factors fit in HBM and exactness is cheaper than
    the estimators' variance.  The data batch is consumed in chunks of ``x_chunk`` examples."""
    from .ggn import gram_from_factor
    N = full_set_size or Z.shape[0]
    M_, Kb = Z.shape[0], X.shape[0]
    beta, gamma = N / M_, N / Kb
    engz, Wm = _factor_of(state, Z, model_type)
    dev, D = Wm.device, engz.D
    d = Wm.shape[0]
    Gd = gram_from_factor(Wm)
    Gd = 0.5 * (Gd + Gd.T)
    lam, U = torch.linalg.eigh(Gd)
    lam = lam.clamp_min(0.0)
    cdiag = 1.0 / (alpha + beta * lam)                                      # C = (alpha I + beta Gd)^-1
    C = (U * cdiag) @ U.T
    Mrow = torch.zeros_like(Wm)
    H = torch.zeros(d, d, device=dev, dtype=torch.float64)                  # Gxz^T Gxz
    tr_x = 0.0
    step = x_chunk or Kb
    for s0 in range(0, Kb, step):
        Xc = X[s0:s0 + step]
        _, Wx = _factor_of(state, Xc, model_type)
        tr_x += float((Wx.double() ** 2).sum()) if _with_constants else 0.0
        Gxz = _gram64(Wx, Wm)                                               # (dx_c, d)
        H += Gxz.T @ Gxz
        A2 = (gamma / alpha) * (Gxz @ C)                                    # (dx_c, d)
        Mrow.addmm_(A2.T.float(), Wx, beta=1.0, alpha=-1.0)
        del Wx
    A1 = (U * (beta * lam * cdiag * cdiag)) @ U.T + (gamma * beta / alpha) * (C @ H @ C)
    Mrow.addmm_(A1.float(), Wm, beta=1.0, alpha=1.0)
    gZ = 2.0 * beta * _input_grad_of_pairing(state, Z, Mrow, model_type)
    # exact value, the reference's scalable_exact formula (:60-84)
    a_inv = 1.0 / alpha
    logdet_term = torch.log1p(beta * a_inv * lam).sum() + D * math.log(alpha)
    Minv = (U * (1.0 / (1.0 / beta + a_inv * lam))) @ U.T                   # (beta^-1 I + alpha^-1 Gd)^-1
    trace1 = (Minv * Gd).sum()
    trace2 = (Minv * H).sum()
    loss = float(logdet_term - a_inv * trace1 - gamma * a_inv ** 2 * trace2)
    if _with_constants:                       # the two Z-independent terms the exact twin drops (:69, :80-82)
        loss += D + gamma * a_inv * tr_x
    return loss, gZ.reshape(Z.shape).to(Z.dtype)


def variational_grad_dense(Z, X, state, alpha, key=None, model_type="classifier", full_set_size=None, **kw):
    """``jax.value_and_grad(alternative_objective_dense)`` (``src/train_inducing.py:195``).  In that function ``S`` and
    ``S_z`` are the PRECISIONS returned by ``compute_curvature_approx_dense`` (``:181-183``), so the value is
    tr(P P_z^-1) + log det P_z — the same function of Z as the scalable objective plus the two constants that one
    drops.  Nothing D x D is formed here: same factor algebra, constants added back."""
    return variational_grad_exact(Z, X, state, alpha, key=key, model_type=model_type, full_set_size=full_set_size,
                                  _with_constants=True, **kw)


def optimize_step(Z, X, map_model_state, alpha, opt_state, rng, zoptimizer, num_mc_samples=None, model_type="classifier",
                  full_set_size=None, scalable=True, **kw):
    """``src/train_inducing.py:199-232``: one optimiser step on Z.  ``zoptimizer`` is any object with
    ``update(grads, opt_state, params) -> (updates, new_opt_state)`` (the optax protocol; :class:`AdamW` below)."""
    fn = variational_grad_scalable if scalable else variational_grad_dense
    loss, grads = fn(Z, X, map_model_state, alpha, key=rng, model_type=model_type, full_set_size=full_set_size, **kw)
    updates, new_opt_state = zoptimizer.update(grads, opt_state, Z)
    return Z + updates, new_opt_state, loss


class AdamW:
    """optax.adamw(lr, weight_decay) as used by the reference's inducing-point scripts."""

    def __init__(self, lr=1e-2, b1=0.9, b2=0.999, eps=1e-8, weight_decay=1e-4):
        self.lr, self.b1, self.b2, self.eps, self.wd = lr, b1, b2, eps, weight_decay

    def init(self, params):
        return dict(t=0, m=torch.zeros_like(params), v=torch.zeros_like(params))

    def update(self, grads, st, params):
        t = st["t"] + 1
        m = self.b1 * st["m"] + (1 - self.b1) * grads
        v = self.b2 * st["v"] + (1 - self.b2) * grads * grads
        mh, vh = m / (1 - self.b1 ** t), v / (1 - self.b2 ** t)
        upd = -self.lr * (mh / (vh.sqrt() + self.eps) + self.wd * params)
        return upd, dict(t=t, m=m, v=v)


def train_inducing_points(map_model_state, zinit, zoptimizer, batches, model_type, rng=0, num_mc_samples=None, alpha=1.0,
                          num_steps=100, full_set_size=None, scalable=True, **kw):
    """``src/train_inducing.py:235-...`` without the plotting: ``batches`` is an iterable (re-iterated when exhausted)
    of data batches X (or (X, y) pairs).  Returns ``(Z, losses)``."""
    z = zinit.clone()
    opt_state = zoptimizer.init(z)
    it = iter(batches)
    losses = []
    for step in range(num_steps):
        try:
            b = next(it)
        except StopIteration:
            it = iter(batches)
            b = next(it)
        Xb = b[0] if isinstance(b, (tuple, list)) else b
        z, opt_state, loss = optimize_step(z, Xb, map_model_state, alpha, opt_state, rng + step, zoptimizer, num_mc_samples,
                                           model_type, full_set_size, scalable, **kw)
        losses.append(loss)
    return z, losses
