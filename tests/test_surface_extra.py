"""Coverage of the remaining SURVEY §8(a) rows on both implementations, plus full-size parity cases of the
BASELINE configs (MNIST-MLP at D = 1 494 154; CIFAR ResNet1M at D = 1 084 586) on the GPU."""
import math

import pytest
import torch

from fixtures import (classification_2d_data, classifier_state, regression_1d_data, sine_data,  # noqa: F401
                      small_model_state, toyregressor_state)
from impl import cpu64, impl  # noqa: F401
import oracle.ggn as og
import oracle.lla as olla


def test_build_WTWz_and_blockwise(impl, classification_2d_data, classifier_state):
    """build_WTWz (src/ggn.py:233) == W^T W_z, and the blockwise per-point factors (src/ggn.py:81-82)."""
    X, y = classification_2d_data
    Xa, Xz = X[::20], X[5::40]                      # 10 data points, 5 "inducing" points
    st = impl.state(classifier_state)
    W, WT = impl.ggn.compute_W_vps(st, impl.tensor(Xa), "classifier")
    Wz, WzT = impl.ggn.compute_W_vps(st, impl.tensor(Xz), "classifier")
    G = impl.ggn.build_WTWz(WT, Wz, (5, 2), d=20, dtype=impl.dtype, block=3)
    Wo, WTo = og.compute_W_vps(classifier_state, Xa, "classifier")
    Wzo, _ = og.compute_W_vps(classifier_state, Xz, "classifier")
    ref = og.build_WTWz(WTo, Wzo, (5, 2), d=20, dtype=torch.float64)
    assert torch.allclose(cpu64(G), ref, rtol=impl.tol(1e-10, 2e-4), atol=impl.tol(1e-12, 2e-4 * ref.abs().max().item()))
    Wb, WTb = impl.ggn.compute_W_vps(st, impl.tensor(Xa), "classifier", blockwise=True)
    Wbo, WTbo = og.compute_W_vps(classifier_state, Xa, "classifier", blockwise=True)
    v = torch.randn(354, dtype=torch.float64, generator=torch.Generator().manual_seed(0))
    u = torch.randn(2, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    assert torch.allclose(cpu64(WTb(3, impl.tensor(v))), WTbo(3, v), rtol=impl.tol(1e-10, 2e-4), atol=impl.tol(1e-12, 1e-5))
    assert torch.allclose(cpu64(Wb(3, impl.tensor(u))), Wbo(3, u), rtol=impl.tol(1e-10, 2e-4), atol=impl.tol(1e-12, 1e-5))


def test_materialize_covariance_and_la_samples(impl, small_model_state, regression_1d_data):
    """materialize_covariance (src/lla.py:160) on an explicit operator; predict_la_samples_dense (:84) shapes
    and moments on the linear model (where LA == LLA)."""
    A = torch.tensor([[2.0, 0.5, 0.0, 0.1], [0.5, 1.0, 0.2, 0.0], [0.0, 0.2, 3.0, 0.3], [0.1, 0.0, 0.3, 1.5]], dtype=torch.float64)
    op = lambda e: (A @ e).reshape(2, 2)
    assert torch.allclose(impl.lla.materialize_covariance(op, 2, 2, "full"), A)
    assert torch.allclose(impl.lla.materialize_covariance(op, 2, 2, "diag").reshape(-1), torch.diagonal(A))
    with pytest.raises(ValueError):
        impl.lla.materialize_covariance(op, 2, 2, "banana")
    if impl.is_hip:
        X, y = regression_1d_data
        st, Xd = impl.state(small_model_state), impl.tensor(X)
        xnew = impl.tensor(torch.tensor([[-0.5], [0.5], [2.0]], dtype=torch.float64))
        fs = cpu64(impl.lla.predict_la_samples_dense(st, xnew, Xd, "regressor", alpha=1.0, num_mc_samples=4000, key=3))
        assert fs.shape == (4000, 3)
        ref = olla.predict_lla_dense(small_model_state, cpu64(xnew), X, "regressor", 1.0)
        assert torch.allclose(fs.mean(0), ref.mean(), atol=0.05)
        assert torch.allclose(fs.var(0), torch.diagonal(ref.covariance()), rtol=0.15)


@pytest.mark.gpu
def test_sample_dense_and_both(sine_data, toyregressor_state):
    """sample_dense / sample_both / inv_matsqrt_dense (src/sample.py:16-52,159-178): the dense twin agrees with
    the matrix-free operator on the same noise, and sample_dense adds theta_MAP."""
    import src.sample as hs
    from lip_amd.utils import flatten_nn_params
    X, y = sine_data
    X = X[::3]                                        # 6 points: a well-conditioned Gram
    st = toyregressor_state.to(device="cuda", dtype=torch.float32)
    Xd = X.cuda().float()
    mf, dense = hs.sample_both(st, Xd, 241, 0.5, 11, "regressor", num_samples=6)
    assert torch.allclose(mf, dense, rtol=2e-3, atol=2e-3 * dense.abs().max().item())
    sd = hs.sample_dense(st, Xd, 241, 0.5, 11, "regressor", num_samples=6)
    flat, _ = flatten_nn_params(st.params)
    assert torch.allclose(sd - flat.cuda(), dense, rtol=1e-4, atol=1e-4)


def test_hutchpp_sampler_variant(impl):
    """hutchpp(Xfun, sampler) (src/stochtrace.py:82-111): exact when the first half of the probes spans R^n."""
    n = 24
    g = torch.Generator().manual_seed(0)
    B = torch.randn(n, n, dtype=torch.float64, generator=g)
    A = impl.tensor(B @ B.T)
    eps = impl.tensor(torch.randn(2 * n + 8, n, dtype=torch.float64, generator=g))
    if impl.is_hip:
        from lip_amd.ggn import BlockOperator
        f = BlockOperator(lambda V: (V @ A.T).contiguous(), (n,), (n,), None, "A")
    else:
        f = lambda v: A @ v
    tr = impl.stochtrace.hutchpp(f, lambda _: eps)
    assert torch.isclose(cpu64(tr), cpu64(torch.trace(A)), rtol=impl.tol(1e-8, 5e-4))


@pytest.mark.gpu
def test_mnist_mlp_full_size_parity():
    """BASELINE configs[2] at full size: LargeClassifier 784-1024-512-256-128-10 (D = 1 494 154), n = 50
    synthetic U[0,1] images, 4 Rademacher probes, alpha = 1e-3: HIP fp32 vs the example-batched float64
    oracle; plus symmetry <u, G v> == <v, G u> as a size-independent property."""
    from lip_amd import krylov
    from lip_amd.scalemodels import LargeClassifier
    from lip_amd.toymodels import create_state
    import src.lla as hl
    net = LargeClassifier((28, 28, 1), [1024, 512, 256, 128], 4, 10)
    st64 = create_state(net, 12345, dtype=torch.float64)
    Z = torch.rand(50, 28, 28, 1, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    vp = hl.compute_curvature_approx(st64.to(device="cuda", dtype=torch.float32), Z.cuda().float(), "classifier", 1e-3,
                                     full_set_size=60000)
    D = vp.engine.D
    assert D == 1_494_154
    V = krylov.fill_rademacher(4, D, 7, "cuda")
    Y = vp(V)
    ref_vp = og.compute_ggn_vp_batched(st64, Z, "classifier", full_set_size=60000)
    ref = torch.stack([ref_vp(v) + 1e-3 * v for v in cpu64(V)[:2]])
    err = (cpu64(Y[:2]) - ref).abs().max() / ref.abs().max()
    assert err < 2e-4, err
    uGv, vGu = (Y[0] * V[1]).sum().item(), (Y[1] * V[0]).sum().item()
    assert abs(uGv - vGu) <= 1e-3 * max(abs(uGv), abs(vGu), 1.0)


@pytest.mark.gpu
def test_cifar_resnet1m_full_size_properties():
    """BASELINE configs[3] at full size (ResNet1M, D = 1 084 586, n = 50, P = 16): one probe against the
    example-batched float64 oracle; linearity, symmetry and positive semi-definiteness for the block."""
    from lip_amd import krylov
    from lip_amd.scalemodels import ResNet1M
    from lip_amd.toymodels import create_state
    import src.ggn as hg
    net = ResNet1M(10)
    st64 = create_state(net, 1231231234, dtype=torch.float64)
    Z = torch.rand(50, 32, 32, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(280300))
    vp = hg.compute_ggn_vp(st64.to(device="cuda", dtype=torch.float32), Z.cuda().float(), "classifier", full_set_size=49000)
    D = vp.engine.D
    assert D == 1_084_586
    V = krylov.fill_rademacher(16, D, 3, "cuda")
    Y = vp(V)
    ref = og.compute_ggn_vp_batched(st64, Z, "classifier", full_set_size=49000)(cpu64(V[0]))
    err = (cpu64(Y[0]) - ref).abs().max() / ref.abs().max()
    assert err < 2e-4, err
    lin = vp((2.0 * V[0] - 3.0 * V[1])[None])[0]
    assert torch.allclose(lin, 2.0 * Y[0] - 3.0 * Y[1], rtol=1e-3, atol=1e-3 * Y.abs().max().item())
    Gm = V @ Y.T                                       # (16, 16) = V G V^T
    assert torch.allclose(Gm, Gm.T, rtol=1e-3, atol=1e-3 * Gm.abs().max().item())
    assert torch.linalg.eigvalsh(0.5 * (Gm + Gm.T).double()).min() > -1e-3 * Gm.abs().max().item()
    # the bench's block size takes other kernels than a 16-probe block (128-row tiles, probe-batched weight gradient,
    # no row split): the first 16 rows of a 256-probe block must agree with the 16-probe block checked above
    V256 = torch.cat([V, krylov.fill_rademacher(240, D, 4, "cuda")])
    Y256 = vp(V256)
    assert (Y256[:16] - Y).abs().max().item() <= 2e-5 * Y.abs().max().item()
    assert torch.isfinite(Y256).all()
    # per-example rows at full size: their sum over examples is the summed product
    eng = vp.engine
    U = torch.randn(4, eng.n, eng.K, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    rows = eng.vjp_rows(U, "l", 1.0)
    assert (rows.sum(1) - eng.vjp(U, "l", 1.0)).abs().max().item() <= 2e-5 * rows.abs().max().item() * eng.n


@pytest.mark.gpu
def test_factor_mode_equals_matrix_free(classification_2d_data, classifier_state):
    """compute_ggn_vp(mode="factor") (materialised W, two GEMMs) == the matrix-free sweep == the oracle."""
    import src.ggn as hg
    X, y = classification_2d_data
    X = X[::10]
    st = classifier_state.to(device="cuda", dtype=torch.float32)
    Xd = X.cuda().float()
    V = torch.randn(5, 354, dtype=torch.float64, generator=torch.Generator().manual_seed(0))
    a = hg.compute_ggn_vp(st, Xd, "classifier", full_set_size=77)(V.cuda().float())
    b = hg.compute_ggn_vp(st, Xd, "classifier", full_set_size=77, mode="factor")(V.cuda().float())
    c = hg.compute_ggn_vp(st, Xd, "classifier", full_set_size=77, mode="auto")(V.cuda().float())
    ref_vp = og.compute_ggn_vp(classifier_state, X, "classifier", full_set_size=77)
    ref = torch.stack([ref_vp(v) for v in V])
    for out in (a, b, c):
        assert torch.allclose(cpu64(out), ref, rtol=2e-4, atol=2e-4 * ref.abs().max().item())


@pytest.mark.gpu
def test_example_chunked_operator(classification_2d_data, classifier_state):
    """Summing per-chunk engines over a split of Z == one engine over all of Z (ragged last chunk)."""
    import src.ggn as hg
    X, y = classification_2d_data
    X = X[::7]                                            # 29 points -> chunks 8, 8, 8, 5
    st = classifier_state.to(device="cuda", dtype=torch.float32)
    V = torch.randn(3, 354, generator=torch.Generator().manual_seed(0)).cuda()
    full = hg.compute_ggn_vp(st, X.cuda().float(), "classifier", full_set_size=300)(V)
    chunked = hg.ExampleChunkedGGN(st, X.cuda().float(), "classifier", full_set_size=300, example_chunk=8)(V)
    assert torch.allclose(full, chunked, rtol=1e-4, atol=1e-4 * full.abs().max().item())


@pytest.mark.gpu
def test_resnet50_real_architecture_properties():
    """BASELINE configs[4] architecture at full width/depth (25.6 M parameters, K = 1000) on 64 x 64 synthetic
    images: linearity, symmetry and PSD of the GGN block, and one probe against the example-batched float64
    oracle (the ImageNet-resolution run only changes the pixel counts)."""
    from lip_amd import krylov
    from lip_amd.scalemodels import ResNet50
    from lip_amd.toymodels import create_state
    import src.ggn as hg
    net = ResNet50(1000, input_shape=(64, 64, 3))
    st64 = create_state(net, 0, dtype=torch.float64)
    Z = torch.randn(3, 64, 64, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    vp = hg.compute_ggn_vp(st64.to(device="cuda", dtype=torch.float32), Z.cuda().float(), "classifier", full_set_size=10000)
    D = vp.engine.D
    assert D == 25_557_032
    V = krylov.fill_rademacher(4, D, 3, "cuda")
    Y = vp(V)
    assert torch.isfinite(Y).all()
    Gm = V @ Y.T
    assert torch.allclose(Gm, Gm.T, rtol=2e-3, atol=2e-3 * Gm.abs().max().item())
    assert torch.linalg.eigvalsh(0.5 * (Gm + Gm.T).double()).min() > -2e-3 * Gm.abs().max().item()
    lin = vp((V[0] - 2.0 * V[1])[None])[0]
    assert torch.allclose(lin, Y[0] - 2.0 * Y[1], rtol=2e-3, atol=2e-3 * Y.abs().max().item())
    ref = og.compute_ggn_vp_batched(st64, Z, "classifier", full_set_size=10000)(cpu64(V[0]))
    err = (cpu64(Y[0]) - ref).abs().max() / ref.abs().max()
    assert err < 5e-4, err


@pytest.mark.gpu
def test_predict_lla_marginals_equal_dense(classification_2d_data, classifier_state, sine_data, toyregressor_state):
    """Closed-form per-point predictive (factor algebra + per-example Jacobian rows) == predict_lla_dense
    (reference src/lla.py:51-82) on the toy problems where the dense D x D route is feasible."""
    import src.lla as hl
    X, y = classification_2d_data
    st = classifier_state.to(device="cuda", dtype=torch.float32)
    Z, Xn = X[:12].cuda().float(), X[12:19].cuda().float()
    a = hl.predict_lla_dense(st, Xn, Z, "classifier", alpha=0.5, full_set_size=200)
    b = hl.predict_lla_marginals(st, Xn, Z, "classifier", alpha=0.5, full_set_size=200, batch=4)
    assert torch.allclose(cpu64(a.mean()), cpu64(b.mean()), atol=1e-5)
    assert (cpu64(a.covariance()) - cpu64(b.covariance())).abs().max() <= 2e-4 * cpu64(a.covariance()).abs().max()
    ref = olla.predict_lla_dense(classifier_state, X[12:19], X[:12], "classifier", 0.5, full_set_size=200)
    assert (cpu64(b.covariance()) - ref.covariance()).abs().max() <= 2e-4 * ref.covariance().abs().max()
    Xs, ys = sine_data
    sr = toyregressor_state.to(device="cuda", dtype=torch.float32)
    a = hl.predict_lla_dense(sr, Xs[10:15].cuda().float(), Xs[:10].cuda().float(), "regressor", alpha=0.5)
    b = hl.predict_lla_marginals(sr, Xs[10:15].cuda().float(), Xs[:10].cuda().float(), "regressor", alpha=0.5)
    assert (cpu64(a.covariance()) - cpu64(b.covariance())).abs().max() <= 2e-4 * cpu64(a.covariance()).abs().max()
