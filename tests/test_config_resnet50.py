"""BASELINE.json configs[4] at its real resolution: the synthetic ImageNet-scale ResNet-50 (25 557 032 parameters,
K = 1000, 224 x 224 x 3 inputs, random init) — not in the reference, so the checks are the ones the
domain offers at a size no dense object exists for:

* one probe against the float64 oracle (example-batched restatement of ``src/ggn.py:133-144``) on ONE image (the
  oracle needs ~20 s of CPU per image at this size);
* symmetry <u, G v> = <v, G u>, positive semi-definiteness, linearity;
* additivity of the data sum — the property the multi-GPU shard and ``ExampleChunkedGGN`` rest on: the operator over
  the images {0, 1, 2} equals the sum of the operators over {0, 1} and {2}, and the chunked operator over 6 images in
  chunks of 4 + 2 equals the single binding.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def r50():
    import lip_amd  # noqa: F401
    from lip_amd.scalemodels import ResNet50
    from lip_amd.toymodels import create_state
    net = ResNet50(1000)
    st64 = create_state(net, seed=1, dtype=torch.float64)
    Z = torch.rand(6, 224, 224, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(3))
    st32 = st64.to(device="cuda", dtype=torch.float32)
    return dict(net=net, st64=st64, st32=st32, Z=Z)


def test_resnet50_224_one_probe_against_oracle(r50):
    from lip_amd.engine import LinearizedNet
    from oracle.ggn import compute_ggn_vp_batched
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    Z1 = r50["Z"][:1]
    eng = LinearizedNet(r50["st32"], Z1.cuda().float(), "classifier", workspace_bytes=4 << 30, max_chunk=2)
    assert eng.D == 25557032 and eng.K == 1000
    v = torch.randn(eng.D, dtype=torch.float64, generator=torch.Generator().manual_seed(4))
    ref = compute_ggn_vp_batched(r50["st64"], Z1, "classifier", full_set_size=10000)(v)
    out = eng.ggn_vp(v.cuda().float()[None], 10000.0, 0.0)[0].double().cpu()
    err = (out - ref).abs().max().item() / ref.abs().max().item()
    # float32 sweep through 53 convolution layers at 224 x 224 against float64: measured 1e-5; bound 3e-4
    assert err <= 3e-4, err


def test_resnet50_224_operator_properties(r50):
    from lip_amd import krylov
    from lip_amd.engine import LinearizedNet
    from lip_amd.ggn import ExampleChunkedGGN
    Zc = r50["Z"].cuda().float()
    st = r50["st32"]
    work = torch.empty(6 << 28, device="cuda")                        # 6 GiB of shared probe workspace
    e012 = LinearizedNet(st, Zc[:3], "classifier", work=work, max_chunk=4)
    D = e012.D
    U = krylov.fill_rademacher(2, D, 5, "cuda")
    V = krylov.fill_normal(2, D, 6, "cuda")
    GU, GV = e012.ggn_vp(U, 1.0, 0.0).double(), e012.ggn_vp(V, 1.0, 0.0).double()
    assert torch.isfinite(GU).all() and torch.isfinite(GV).all()
    uGv, vGu = (U.double() * GV).sum(1), (V.double() * GU).sum(1)
    scale = (GU.norm(dim=1) * V.double().norm(dim=1))
    assert ((uGv - vGu).abs() / scale).max().item() <= 1e-5                       # symmetry
    assert ((V.double() * GV).sum(1) >= 0).all() and ((U.double() * GU).sum(1) >= 0).all()      # PSD
    a, b = 0.7, -1.3
    lin = e012.ggn_vp((a * U + b * V).contiguous(), 1.0, 0.0).double()
    assert (lin - (a * GU + b * GV)).abs().max().item() <= 2e-5 * (a * GU + b * GV).abs().max().item()   # linearity
    # additivity over examples
    e01 = LinearizedNet(st, Zc[:2], "classifier", work=work, max_chunk=4)
    e2 = LinearizedNet(st, Zc[2:3], "classifier", work=work, max_chunk=4)
    parts = e01.ggn_vp(V, 1.0, 0.0).double() + e2.ggn_vp(V, 1.0, 0.0).double()
    assert (parts - GV).abs().max().item() <= 2e-5 * GV.abs().max().item()
    del e01, e2, e012
    # the chunked operator (one shared workspace, 4 + 2 images) against single bindings of the same images
    ch = ExampleChunkedGGN(st, Zc, "classifier", full_set_size=10000, example_chunk=4, workspace_bytes=6 << 30, max_probes=2)
    assert len(ch.engines) == 2 and ch.engines[0].work is ch.engines[1].work
    Yc = ch(V, alpha=0.25).double()
    e_all = LinearizedNet(st, Zc, "classifier", work=work, max_chunk=2)
    Ya = e_all.ggn_vp(V, 10000.0 / 6.0, 0.25).double()
    assert (Yc - Ya).abs().max().item() <= 2e-5 * Ya.abs().max().item()


def test_resnet50_224_log_marginal_likelihood_in_alpha(r50):
    """configs[4]'s log-marginal-likelihood optimisation (``src/train_alpha.py:13-59``) at 224 x 224 with one inducing
    image (d = K = 1000; the 102 GB factor is not materialised — the Gram is assembled matrix-free).  No dense object
    exists to compare with at D = 25.6 M, so: the analytic d/d(log alpha) against a central difference of the value,
    the Gram's spectrum is PSD with rank <= K - 1 = 999 (softmax: L sqrt(p) = 0), and Adam on log alpha ascends."""
    from lip_amd.ggn import clear_engine_cache
    from lip_amd.train_alpha import _lml_from_spectrum, _spectrum, fit_alpha
    st, Z1 = r50["st32"], r50["Z"][:1].cuda().float()
    lam, D, theta2 = _spectrum(Z1, st, "classifier")
    assert D == 25557032 and lam.numel() == 1000
    assert lam.min().item() >= 0.0 and (lam > 1e-5 * lam.max()).sum().item() <= 999
    v0, g0 = _lml_from_spectrum(2.0, lam, D, theta2, 10000.0)
    h = 1e-4
    vp = _lml_from_spectrum(2.0 * (1 + h), lam, D, theta2, 10000.0)[0]
    vm = _lml_from_spectrum(2.0 * (1 - h), lam, D, theta2, 10000.0)[0]
    fd = (vp - vm) / (torch.log(torch.tensor((1 + h) / (1 - h), dtype=torch.float64)).item())
    assert abs(fd - g0) <= 1e-5 * max(1.0, abs(g0))
    a_fit, hist = fit_alpha(Z1, st, "classifier", full_set_size=10000, alpha0=1.0, steps=50)
    assert hist[-1][1] > hist[0][1] and a_fit > 0
    clear_engine_cache()
