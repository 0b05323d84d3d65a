"""Winograd F(2x2, 3x3) route of the 3x3 / stride-1 layers (csrc/lip_mfma.hip: igemm_wino_kernel):
the same products with the route forced on every eligible launch and switched off.  The two differ by the rounding of
the input / weight / output transforms (f32 in, f32 accumulate on both routes), so they agree to a few float32 ulps of
the result's scale and are NOT bit-identical — which proves the Winograd launches are the ones taken; both are checked
against the float64 oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _binding(n, hw=32, seed=3):
    from lip_amd.engine import LinearizedNet
    from lip_amd.scalemodels import ResNet1M
    from lip_amd.toymodels import create_state
    net = ResNet1M(10, input_shape=(hw, hw, 3))
    st = create_state(net, seed=seed, dtype=torch.float32)
    Z = torch.rand(n, hw, hw, 3, generator=torch.Generator().manual_seed(5)).cuda()
    return st, Z, LinearizedNet(st, Z, "classifier", workspace_bytes=1 << 30, max_chunk=8)


@pytest.mark.parametrize("n,P", [(2, 3), (5, 8)])
def test_winograd_matches_the_direct_kernels_and_is_active(n, P):
    from lip_amd import _native as nv, krylov
    lib = nv.load()
    st, Z, eng = _binding(n)
    V = krylov.fill_rademacher(P, eng.D, 11, "cuda")
    before = lib.lip_get_winograd()
    try:
        assert lib.lip_set_winograd(0) == 0
        y_off = eng.ggn_vp(V, 1.0, 0.0).clone()
        u_off = eng.jvp(V).clone()
        assert lib.lip_set_winograd(2) == 0
        y_on = eng.ggn_vp(V, 1.0, 0.0).clone()
        u_on = eng.jvp(V).clone()
    finally:
        lib.lip_set_winograd(before)
    assert torch.isfinite(y_on).all() and torch.isfinite(u_on).all()
    for on, off, what in ((y_on, y_off, "ggn_vp"), (u_on, u_off, "jvp")):
        scale = off.abs().max().item()
        err = (on - off).abs().max().item() / scale
        # 20 layers of ~2e-7 per-layer transform rounding: measured ~1e-6; bound 2e-5
        assert err <= 2e-5, (what, err)
        assert not torch.equal(on, off), f"{what}: Winograd launches were not taken (bit-identical results)"


def test_winograd_against_the_float64_oracle():
    """forced Winograd vs the example-batched float64 oracle on a 6-image binding: the tolerance the direct route is
    held to in tests/test_surface_extra.py"""
    from lip_amd import _native as nv, krylov
    from lip_amd.scalemodels import ResNet1M
    from lip_amd.toymodels import create_state
    import oracle.ggn as og
    import src.ggn as hg
    lib = nv.load()
    st64 = create_state(ResNet1M(10), 1231231234, dtype=torch.float64)      # seeds of tests/test_surface_extra.py
    Z = torch.rand(50, 32, 32, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(280300))[:6]
    vp = hg.compute_ggn_vp(st64.to(device="cuda", dtype=torch.float32), Z.cuda().float(), "classifier", full_set_size=300)
    V = krylov.fill_rademacher(2, vp.engine.D, 7, "cuda")
    before = lib.lip_get_winograd()
    try:
        lib.lip_set_winograd(2)
        y = vp(V).double().cpu()
        lib.lip_set_winograd(0)
        y_direct = vp(V).double().cpu()
    finally:
        lib.lip_set_winograd(before)
    ref_vp = og.compute_ggn_vp_batched(st64, Z, "classifier", full_set_size=300)
    ref = torch.stack([ref_vp(v) for v in V.double().cpu()])
    err = ((y - ref).abs().max() / ref.abs().max()).item()
    err_direct = ((y_direct - ref).abs().max() / ref.abs().max()).item()
    assert err < 2e-4, (err, err_direct)
    assert err <= max(10.0 * err_direct, 2e-5), (err, err_direct)


@pytest.mark.parametrize("hw", [28, 24])
def test_ragged_tile_blocks_and_tile_rows(hw):
    """maps whose tile grid is not a multiple of the block rectangle (14 x 14, 7 x 7, 12 x 12, 6 x 6, 3 x 3 tiles): masked
    tiles of the implicit GEMM, the per-tile coordinate path of the weight gradient (tile rows not a multiple of 4 tiles
    wide); 7 x 7 maps (hw = 28) are ineligible and fall through to the direct kernels"""
    from lip_amd import _native as nv, krylov
    lib = nv.load()
    st, Z, eng = _binding(3, hw=hw)
    V = krylov.fill_rademacher(3, eng.D, 13, "cuda")
    before = lib.lip_get_winograd()
    try:
        lib.lip_set_winograd(0)
        y_off = eng.ggn_vp(V, 1.0, 0.5).clone()
        lib.lip_set_winograd(2)
        y_on = eng.ggn_vp(V, 1.0, 0.5).clone()
    finally:
        lib.lip_set_winograd(before)
    err = (y_on - y_off).abs().max().item() / y_off.abs().max().item()
    assert err <= 2e-5, err
    assert not torch.equal(y_on, y_off)


def test_the_primal_tape_never_takes_the_route():
    """ReLU gates and pooling arg-maxima are taken from the direct sums whatever the switch says: the cached primal
    values of a binding are bit-identical with the route forced and off"""
    from lip_amd import _native as nv
    lib = nv.load()
    before = lib.lip_get_winograd()
    try:
        lib.lip_set_winograd(0)
        prim_off = _binding(4)[2].prim.clone()
        lib.lip_set_winograd(2)
        prim_on = _binding(4)[2].prim.clone()
    finally:
        lib.lip_set_winograd(before)
    assert torch.equal(prim_on, prim_off)
