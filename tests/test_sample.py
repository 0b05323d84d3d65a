"""Restatement of the reference's ``tests/test_sample.py`` (runnable subset) plus HIP-vs-oracle
parity of the sampler on identical noise.  MAP weights are seeded random inits (checkpoints absent)."""
import math

import pytest
import torch

from fixtures import (classification_2d_data, classifier_state, regression_1d_data, sine_data,  # noqa: F401
                      small_model_state, toyregressor_state)
from impl import cpu64, impl  # noqa: F401
from lip_amd.utils import flatten_nn_params
import oracle.ggn as og
import oracle.sample as osamp
from oracle.matfree import dense_funm_sym_eigh, funm_lanczos_sym, tridiag_sym


def test_WT_W_vps(impl, regression_1d_data, small_model_state):
    """reference :19-49: W(W^T(I)) == dense GGN, atol 1e-8 (fp32 HIP: 1e-5)."""
    X, y = regression_1d_data
    st, Xd = impl.state(small_model_state), impl.tensor(X)
    D = 2
    I = torch.eye(D, dtype=impl.dtype, device=impl.device)
    full_GGN, *_ = og.compute_ggn_dense(small_model_state, X, "regressor")
    Wfun, WTfun = impl.ggn.compute_W_vps(st, Xd, "regressor")
    WT_out = impl.rows(WTfun, I)
    composite = impl.rows(Wfun, WT_out)
    assert torch.all(torch.isclose(cpu64(composite), full_GGN, atol=impl.tol(1e-8, 1e-5))), "GGNs don't match!"


def test_WT_W_vps_2(impl, classification_2d_data, classifier_state):
    """reference :51-105 (classifier, float64 there): W(W^T(I)) == dense GGN and the Gram's log-det."""
    X, y = classification_2d_data
    X = X[::8]
    st, Xd = impl.state(classifier_state), impl.tensor(X)
    flat, _ = flatten_nn_params(classifier_state.params)
    D = flat.shape[0]
    full_GGN, *_ = og.compute_ggn_dense(classifier_state, X, "classifier")
    Wfun, WTfun = impl.ggn.compute_W_vps(st, Xd, "classifier")
    cols = torch.arange(0, D, 1 if impl.is_hip else 37)
    I = torch.eye(D, dtype=impl.dtype, device=impl.device)[cols]
    composite = impl.rows(Wfun, impl.rows(WTfun, I))
    assert torch.all(torch.isclose(cpu64(composite), full_GGN[cols], atol=impl.tol(1e-8, 2e-5))), "GGNs don't match!"
    alpha = 0.37
    dummy = WTfun(torch.zeros(D, dtype=impl.dtype, device=impl.device))
    d = dummy.numel()
    WTW = impl.ggn.build_WTW(Wfun, WTfun, tuple(dummy.shape), d, dtype=impl.dtype, block=1)
    _, logdet = torch.linalg.slogdet(torch.eye(d, dtype=torch.float64) + cpu64(WTW) / alpha)
    _, logdet_ref = torch.linalg.slogdet(torch.eye(D, dtype=torch.float64) + full_GGN / alpha)
    assert torch.isclose(logdet, logdet_ref, rtol=impl.tol(1e-8, 1e-4))      # Sylvester: det(I + W^T W/a) = det(I + W W^T/a)
    assert torch.allclose(cpu64(WTW), cpu64(WTW).T)


@pytest.mark.gpu
def test_build_WTW_streamed_routes_agree(classification_2d_data, classifier_state):
    """build_WTW (src/ggn.py:198-227) has three routes on the engine: the materialised factor, column blocks streamed
    through per-example backward sweeps (what a factor too large for HBM takes: ResNet-50, K = 1000) and column blocks
    of one-hot cotangents through the summed vjp.  Same Gram from all three, and equal to the oracle's."""
    from impl import _Hip
    impl = _Hip()
    X, _ = classification_2d_data
    X = X[::8]
    st, Xd = impl.state(classifier_state), impl.tensor(X)
    W, WT = impl.ggn.compute_W_vps(st, Xd, "classifier", full_set_size=77)
    inner = WT.out_shape
    d = int(torch.tensor(inner).prod())
    G_fac = impl.ggn.build_WTW(W, WT, inner, d, dtype=torch.float64, block=1)
    G_rows = impl.ggn.build_WTW(W, WT, inner, d, dtype=torch.float64, block=3, factor_bytes_limit=0)
    scale = W.factor_scale
    del W.factor_scale                                    # without the scale: the one-hot route
    G_hot = impl.ggn.build_WTW(W, WT, inner, d, dtype=torch.float64, block=3, factor_bytes_limit=0)
    W.factor_scale = scale
    Wo, WTo = og.compute_W_vps(classifier_state, X, "classifier", full_set_size=77)
    ref = og.build_WTW(Wo, WTo, inner, d, dtype=torch.float64)
    tol = 2e-5 * float(ref.abs().max())
    for G in (G_fac, G_rows, G_hot):
        assert float((cpu64(G) - ref).abs().max()) < tol


def test_nullproj(impl, sine_data, toyregressor_state):
    """reference :110-152: v - W (W^T W)^-1 W^T v lies in the kernel of the GGN (CG inverse, atol 1.5e-3)."""
    X, y = sine_data
    st, Xd = impl.state(toyregressor_state), impl.tensor(X)
    D = 241
    v = impl.tensor(torch.randn(D, dtype=torch.float64, generator=torch.Generator().manual_seed(41234)) * 10)
    Wfun, WTfun = impl.ggn.compute_W_vps(st, Xd, "regressor")
    if impl.is_hip:
        from lip_amd import krylov
        comp = lambda U: WTfun.rows(Wfun.rows(U))
        x, info = krylov.cg(comp, WTfun(v)[None].contiguous())
        full_out = v - Wfun(x[0])
    else:
        from oracle.matfree import cg
        x, _ = cg(lambda u: WTfun(Wfun(u)), WTfun(v))
        full_out = v - Wfun(x)
    assert full_out.shape == (D,)
    resid = cpu64(Wfun(WTfun(full_out)))
    scale = cpu64(Wfun(WTfun(v))).abs().max()
    # reference: atol 1.5e-3 in float64.  The Gram of these 16 neighbouring sine points has condition number ~1e17, ten
    # decades beyond 1/eps_f32: the literal route — a float32 CG recurrence on float32 products — is CHAOTIC there, not
    # merely less accurate.  scripts/nullproj_probe.py (round 3, MI355X): the residual relative to the scale ranges
    # over 3e-5 ... 9e-1 across six right-hand sides x six iteration caps, moves 4.8e-3 -> 3.4e-2 under a 1e-5
    # relative perturbation of v, and changes by 10x between two builds that agree bit for bit on every block product
    # (round 2 asserted 5e-3 here: the measured value was 4.8e-3, luck).  So for the float32 path only finiteness is
    # asserted on the literal route, and the reference's 1.5e-3 is asserted on the route the product actually uses
    # for this projector (orthonormalised factor, float64 coefficients) — below.  The float64 oracle keeps the
    # reference's assertion unchanged.
    if impl.is_hip:
        assert bool(torch.isfinite(resid).all())
    else:
        tol = 1.5e-3 * max(1.0, scale.item())
        assert torch.all(resid.abs() <= tol), f"full_out should be in the kernel of the GGN: {resid.abs().max()} > {tol}"
    if impl.is_hip:
        from lip_amd import krylov
        Qm = impl.sample.inv_matsqrt_vp(st, Xd, D, 0.5, "regressor").parts.Qm
        c = krylov.dot_nt(v[None].contiguous(), Qm)                      # <q_k, v> in float64
        proj = krylov.rows_combine(-c, Qm, Z=v[None].contiguous(), zscale=1.0)[0]      # v - Q^T Q v
        resid2 = cpu64(Wfun(WTfun(proj)))
        assert torch.all(resid2.abs() <= 1.5e-3 * max(1.0, scale.item())), resid2.abs().max()


def test_matfree_invsqrt(impl):
    """reference :334-355: Lanczos inverse square root on diag(1..100)/100, 20 steps, ones vector, rtol 1e-1
    (un-patched eigh).  RNG-free."""
    D = 100
    matdiag = torch.arange(1, D + 1, dtype=torch.float64) / D
    res = torch.ones(D, dtype=torch.float64) / torch.sqrt(matdiag)
    if impl.is_hip:
        from lip_amd import krylov
        md = impl.tensor(matdiag)
        f = krylov.funm_lanczos_sym(krylov.dense_funm_sym_eigh(lambda x: 1.0 / torch.sqrt(x)), 20)
        mf_res = f(lambda V: V * md, torch.ones(3, D, device="cuda"))[1]
    else:
        f = funm_lanczos_sym(dense_funm_sym_eigh(lambda x: 1.0 / torch.sqrt(x)), tridiag_sym(20))
        mf_res = f(lambda v: matdiag * v, torch.ones(D, dtype=torch.float64))
    assert torch.all(torch.isclose(res, cpu64(mf_res), rtol=1e-1)), "hmm"


def test_lanczos_breakdown_is_finite(impl):
    """A rank-2-plus-identity matrix exhausts its Krylov space after 3 steps; 8 requested steps must still
    give the exact f(A) b (the reference's matfree has no guard there)."""
    n = 12
    g = torch.Generator().manual_seed(0)
    B = torch.randn(n, 2, dtype=torch.float64, generator=g)
    A = 0.5 * torch.eye(n, dtype=torch.float64) + B @ B.T
    b = torch.randn(n, dtype=torch.float64, generator=g)
    ev, U = torch.linalg.eigh(A)
    ref = (U * ev.rsqrt()) @ U.T @ b
    if impl.is_hip:
        from lip_amd import krylov
        Ad = impl.tensor(A)
        f = krylov.funm_lanczos_sym(krylov.dense_funm_sym_eigh(lambda x: 1.0 / torch.sqrt(x)), 8)
        out = f(lambda V: (V @ Ad).contiguous(), impl.tensor(b)[None].contiguous())[0]
    else:
        f = funm_lanczos_sym(dense_funm_sym_eigh(lambda x: 1.0 / torch.sqrt(x)), tridiag_sym(8))
        out = f(lambda v: A @ v, b)
    assert torch.isfinite(cpu64(out)).all()
    assert torch.allclose(cpu64(out), ref, rtol=impl.tol(1e-8, 2e-3), atol=impl.tol(1e-10, 2e-3))


@pytest.mark.parametrize("clip_min", [None, 1.0])
def test_inv_matsqrt_vp_matches_dense(impl, sine_data, toyregressor_state, clip_min):
    """Matrix-free A^(-1/2) v against the exact dense (GGN + alpha I)^(-1/2) v and, for the HIP path, against
    the oracle's matrix-free operator with the same settings.  The Gram W^T W of the 16 neighbouring sine
    points has condition number ~1e17, so the reference's literal ``solve`` (src/sample.py:81,135) is
    ill-posed here; both sides use the truncated pseudo-inverse (``gram_rtol`` = 1e-6).  ``clip_min=1.0`` is
    the reference's monkey-patched clip (src/matfree_monkeypatch.py:19) — "parity unpinned" for it."""
    X, y = sine_data
    alpha = 0.5
    D = 241
    g = torch.Generator().manual_seed(484)
    V = torch.randn(3, D, dtype=torch.float64, generator=g)
    ref_fun = osamp.inv_matsqrt_vp(toyregressor_state, X, D, alpha, "regressor", clip_min=clip_min,
                                   gram_rtol=osamp.PRODUCT_GRAM_RTOL)
    ref = torch.stack([ref_fun(v) for v in V])
    if clip_min is None:
        GGN, *_ = og.compute_ggn_dense(toyregressor_state, X, "regressor")
        ev, U = torch.linalg.eigh(GGN + alpha * torch.eye(D, dtype=torch.float64))
        exact = V @ ((U * ev.rsqrt()) @ U.T)
        assert torch.allclose(ref, exact, rtol=1e-3, atol=1e-3 * exact.abs().max().item())
    if impl.is_hip:
        for method in ("lanczos", "eigh"):
            fun = impl.sample.inv_matsqrt_vp(impl.state(toyregressor_state), impl.tensor(X), D, alpha, "regressor",
                                             clip_min=clip_min, method=method)
            out = cpu64(fun.rows(impl.tensor(V)))
            assert torch.allclose(out, ref, rtol=3e-3, atol=3e-3 * ref.abs().max().item()), method


def test_sample_fun_tiny(impl, regression_1d_data, small_model_state):
    """reference :467-479 (1000 samples; compares the alpha = 0.5 sampler with the dense posterior at
    alpha = 1/sqrt(0.5), SURVEY §4.1-8 — kept, atol 1e-1) plus the correct-alpha comparison."""
    X, y = regression_1d_data
    st, Xd = impl.state(small_model_state), impl.tensor(X)
    alpha = 0.5
    post_ref = impl.lla.posterior_lla_dense(st, Xd, alpha=1.0 / alpha ** 0.5, model_type="regressor")
    post_true = impl.lla.posterior_lla_dense(st, Xd, alpha=alpha, model_type="regressor")
    S = 1000 if impl.is_hip else 300
    kw = {} if impl.is_hip else dict(gram_rtol=osamp.PRODUCT_GRAM_RTOL, clip_min=None)   # d = 4 > D = 2: singular Gram
    samples = cpu64(impl.sample.sample(st, Xd, 2, alpha=alpha, key=1392, model_type="regressor", num_samples=S, **kw))
    assert torch.isfinite(samples).all()
    # the reference compares the zero-mean samples with theta_MAP (passes there only because its
    # |theta| <~ 0.1); the sampler is zero-mean (src/sample.py:153-154), so compare with 0.
    assert torch.all(torch.isclose(torch.zeros(2, dtype=torch.float64), samples.mean(0), atol=1.1e-1)), "Means are not close!"
    assert torch.all(torch.isclose(cpu64(post_ref.stddev()), samples.std(0), atol=1e-1)), "Stdevs are not close!"
    assert torch.all(torch.isclose(cpu64(post_true.stddev()), samples.std(0), rtol=0.15)), "Stdevs (correct alpha)"


def test_sample_fun_regressor(impl, sine_data, toyregressor_state):
    """reference :482-508: sampler moments vs the dense posterior on the sine net, atol 1e-1 (zero-mean
    sampler: the reference compares against theta_MAP with |theta| <~ 0.1 there; here the mean check is on
    the zero-mean samples themselves)."""
    X, y = sine_data
    st, Xd = impl.state(toyregressor_state), impl.tensor(X)
    alpha = 0.5
    post = impl.lla.posterior_lla_dense(st, Xd, model_type="regressor", alpha=alpha)
    S = 1500 if impl.is_hip else 150
    kw = {} if impl.is_hip else dict(gram_rtol=osamp.PRODUCT_GRAM_RTOL, clip_min=None)
    samples = cpu64(impl.sample.sample(st, Xd, 241, alpha=alpha, key=1392, model_type="regressor", num_samples=S, **kw))
    assert torch.isfinite(samples).all()
    se = 4.0 * cpu64(post.stddev()) / math.sqrt(S)
    assert torch.all(samples.mean(0).abs() <= se + 1e-1), "Means are not close!"
    # reference atol 1e-1 at S = 1500; never tighter than 5 standard errors of a sample stddev at this S
    tol = max(1e-1, 5.0 * cpu64(post.stddev()).max().item() / math.sqrt(2 * S))
    assert torch.all(torch.isclose(cpu64(post.stddev()), samples.std(0), atol=tol)), "Stdevs are not close!"


@pytest.mark.gpu
def test_sample_classifier_hip_matches_oracle(classification_2d_data, classifier_state):
    """Classifier sampling (the reference's tests for it end in ``assert False``: W^T W is singular there,
    SURVEY §4.1-5).  With the pseudo-inverse the operator is well defined: HIP vs the dense A^(-1/2)."""
    import src.sample as hs
    X, y = classification_2d_data
    X = X[::10]
    D = 354
    alpha = 0.5
    st = classifier_state.to(device="cuda", dtype=torch.float32)
    fun = hs.inv_matsqrt_vp(st, X.cuda().float(), D, alpha, "classifier", full_set_size=200, method="eigh")
    V = torch.randn(4, D, dtype=torch.float64, generator=torch.Generator().manual_seed(3))
    out = cpu64(fun.rows(V.cuda().float()))
    GGN, *_ = og.compute_ggn_dense(classifier_state, X, "classifier", full_set_size=200)
    ev, U = torch.linalg.eigh(GGN + alpha * torch.eye(D, dtype=torch.float64))
    ref = V @ ((U * ev.rsqrt()) @ U.T)
    assert torch.allclose(out, ref, rtol=5e-3, atol=5e-3 * ref.abs().max().item())


@pytest.mark.gpu
def test_sample_lanczos_dspace(sine_data, toyregressor_state):
    """D-space Lanczos sampler (GGN + alpha I)^(-1/2) eps against the dense eigh answer on the same eps."""
    import src.sample as hs
    from lip_amd import krylov
    X, y = sine_data
    alpha, D = 0.5, 241
    st = toyregressor_state.to(device="cuda", dtype=torch.float32)
    S = hs.sample_lanczos(st, X.cuda().float(), D, alpha, 7, "regressor", num_samples=5, num_matvecs=40)
    Eps = krylov.fill_normal(5, D, 7 * 1000003, "cuda")
    GGN, *_ = og.compute_ggn_dense(toyregressor_state, X, "regressor")
    ev, U = torch.linalg.eigh(GGN + alpha * torch.eye(D, dtype=torch.float64))
    ref = cpu64(Eps) @ ((U * ev.rsqrt()) @ U.T)
    assert torch.allclose(cpu64(S), ref, rtol=5e-3, atol=5e-3 * ref.abs().max().item())
