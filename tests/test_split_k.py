"""Split-K implicit GEMM of under-filled launches (csrc/lip_mfma.hip: K-tiles over gridDim.z, raw sums to scratch planes,
igemm_finish_kernel adds them and runs the fused epilogue): the same product with the split on and
off — the two differ only in the summation order of the K axis, so they agree to float32 rounding, and they are NOT
bit-identical, which proves that the split path is the one the few-probe launches take."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P", [1, 2])
def test_split_k_matches_unsplit_and_is_active(P):
    from lip_amd import _native as nv, krylov
    from lip_amd.engine import LinearizedNet
    from lip_amd.scalemodels import ResNet1M
    from lip_amd.toymodels import create_state
    lib = nv.load()
    net = ResNet1M(10)
    st = create_state(net, seed=3, dtype=torch.float32)
    Z = torch.rand(2, 32, 32, 3, generator=torch.Generator().manual_seed(5)).cuda()
    eng = LinearizedNet(st, Z, "classifier", workspace_bytes=1 << 30, max_chunk=4)
    V = krylov.fill_rademacher(P, eng.D, 11, "cuda")
    try:
        assert lib.lip_set_split_k(0) == 0
        y_off = eng.ggn_vp(V, 1.0, 0.0).clone()
        assert lib.lip_set_split_k(1) == 0
        y_on = eng.ggn_vp(V, 1.0, 0.0).clone()
    finally:
        lib.lip_set_split_k(1)
    assert torch.isfinite(y_on).all() and torch.isfinite(y_off).all()
    scale = y_off.abs().max().item()
    err = (y_on - y_off).abs().max().item() / scale
    # reordered float32 sums of <= 1152 terms per output, propagated through 20 layers: measured 2e-7; bound 5e-6
    assert err <= 5e-6, err
    assert not torch.equal(y_on, y_off), "split-K launches were not taken (bit-identical results)"
