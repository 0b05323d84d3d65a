"""GPU parity of the HIP engine against the CPU oracle / tape emulator (fp32 vs float64).

Tolerance: the kernels compute exact-f32 GEMMs (v_mfma_f32_32x32x2_f32 == fmaf chain) with f32
accumulation, so errors are O(1e-7 * sum|a*b|); the tests bound max|err| <= 2e-4 * max|ref|
(and 5e-5 for the small toy nets), stated here per BASELINE.json's "fp32 tolerance".
"""
import math

import pytest
import torch

from lip_amd import _native as nv
from lip_amd.engine import LinearizedNet, build_consts
from lip_amd.scalemodels import LargeClassifier, LeNet5, ResNet1M, ResNet50
from lip_amd.toymodels import SimpleClassifier, SimpleRegressor, create_state
from lip_amd.utils import flatten_nn_params
from tape_emulator import TapeMachine

pytestmark = pytest.mark.gpu
F64 = torch.float64


def _cases():
    g = torch.Generator().manual_seed(0)
    return {
        "sine_regressor": (SimpleRegressor(8, 4), torch.randn(16, 1, dtype=F64, generator=g), "regressor", 7),
        "xor_classifier": (SimpleClassifier(16, 2, 2), torch.randn(32, 2, dtype=F64, generator=g), "classifier", 5),
        "mlp_ragged": (LargeClassifier((6, 6, 1), [40, 24], 2, 5), torch.rand(9, 6, 6, 1, dtype=F64, generator=g),
                       "classifier", 3),
        "mlp_wide": (LargeClassifier((12, 12, 1), [200, 136, 72], 3, 10), torch.rand(50, 12, 12, 1, dtype=F64, generator=g),
                     "classifier", 4),
        "resnet_tiny": (ResNet1M(4, input_shape=(8, 8, 3), widths=(4, 8, 12), blocks_per_stage=2),
                        torch.rand(3, 8, 8, 3, dtype=F64, generator=g), "classifier", 2),
        "resnet_small": (ResNet1M(10, input_shape=(16, 16, 3), widths=(32, 64, 128), blocks_per_stage=1),
                         torch.rand(5, 16, 16, 3, dtype=F64, generator=g), "classifier", 3),
        "resnet50_tiny": (ResNet50(6, input_shape=(20, 20, 3), stem=8, widths=(4, 8), blocks=(2, 1)),
                          torch.rand(2, 20, 20, 3, dtype=F64, generator=g), "classifier", 2),
        "resnet50_small": (ResNet50(100, input_shape=(32, 32, 3), stem=16, widths=(16, 32), blocks=(1, 1)),
                           torch.rand(3, 32, 32, 3, dtype=F64, generator=g), "classifier", 2),
        "lenet5": (LeNet5(10), torch.rand(7, 28, 28, 1, dtype=F64, generator=g), "classifier", 3),
        "resnet_gray": (ResNet1M(3, input_shape=(12, 12, 1), widths=(8, 16), blocks_per_stage=1),
                        torch.rand(4, 12, 12, 1, dtype=F64, generator=g), "classifier", 2),
    }


def _rel(a, b):
    return ((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _localise(eng, tm, V, U, P, mode, c, which_list):
    """Run both machines op by op and report the first op whose outputs differ."""
    msgs = []
    Vd = V.float().cuda().contiguous()
    Yd = torch.zeros(P, eng.D, device="cuda")
    Hd = (U.float().cuda().contiguous() if U is not None else torch.zeros(P, eng.n * eng.K, device="cuda"))
    eng.work.zero_()
    tm.work.zero_()
    tm.V = V.to(F64).contiguous().reshape(-1)
    tm.Y = torch.zeros(P * eng.D, dtype=F64)
    tm.H = (U.to(F64).contiguous().reshape(-1) if U is not None else torch.zeros(P * eng.n * eng.K, dtype=F64))
    for which in which_list:
        for i, op in enumerate(eng.cn.tapes[which]):
            if which == 2 and op.kind == nv.OP_HEAD and mode == nv.HEAD_GGN:
                continue
            nv.check(eng.lib.lip_debug_run_ops(eng.h, which, i, 1, nv.ptr(Vd), nv.ptr(Yd), nv.ptr(Hd), P, mode, c,
                                               nv.stream_ptr()), "debug_run_ops")
            torch.cuda.synchronize()
            tm.run_op(op, P, mode, c)
            for name, dev, ref in (("work", eng.work, tm.work), ("Y", Yd.reshape(-1), tm.Y), ("H", Hd.reshape(-1), tm.H)):
                err = (dev.double().cpu() - ref).abs().max().item()
                scale = ref.abs().max().item() + 1e-30
                if not (err <= 2e-4 * scale):
                    msgs.append(f"tape {which} op {i} kind {op.kind}: {name} max err {err:.3e} (scale {scale:.3e})")
            if msgs:
                return msgs
    return msgs


@pytest.mark.parametrize("name", list(_cases().keys()))
def test_engine_matches_tape_semantics(name):
    net, Z, model_type, P = _cases()[name]
    state = create_state(net, 3, dtype=F64, logvar=-0.3)
    eng = LinearizedNet(state, Z, model_type, workspace_bytes=1 << 28, max_chunk=P)
    torch.cuda.synchronize()
    flat, _ = flatten_nn_params(state.params)
    consts = build_consts(eng.cn, state.params, state.batch_stats, "cpu", F64)
    tm = TapeMachine(eng.cn, flat, consts, Z, chunk=eng.chunk)
    tm.primal()
    tol = 2e-4
    # primal cache
    perr = (eng.prim.double().cpu() - tm.prim).abs().max().item()
    assert perr <= tol * tm.prim.abs().max().item(), f"primal cache differs: {perr}"

    g = torch.Generator().manual_seed(1)
    V = torch.randn(P, eng.D, dtype=F64, generator=g)
    U = torch.randn(P, eng.n, eng.K, dtype=F64, generator=g)
    n = Z.shape[0]
    scale = 2.5 * (math.exp(0.3) if model_type == "regressor" else 1.0)
    Y = eng.ggn_vp(V, scale, 0.37)
    Yr = tm.ggn_vp(V, scale, 0.37)
    Uh = eng.jvp(V, "lt", 1.3)
    Ur = tm.jvp(V, nv.HEAD_LT, 1.3)
    Yw = eng.vjp(U, "l", 0.7)
    Ywr = tm.vjp(U, nv.HEAD_L, 0.7)
    Jr = eng.jvp(V, "raw")
    Jrr = tm.jvp(V, nv.HEAD_OUT, 1.0)
    Jt = eng.vjp(U, "raw")
    Jtr = tm.vjp(U, nv.HEAD_IN, 1.0)
    # per-example rows (lip_vjp_rows): row (p, i) is the summed product of the cotangent masked to example i
    rows = eng.vjp_rows(U, "l", 0.7)
    torch.cuda.synchronize()
    assert rows.shape == (P, n, eng.D)
    errs = dict(ggn=_rel(Y, Yr), wt=_rel(Uh, Ur), w=_rel(Yw, Ywr), jvp=_rel(Jr, Jrr), vjp=_rel(Jt, Jtr),
                rows_sum=_rel(rows.sum(1), Ywr))
    for i in sorted({0, n // 2, n - 1}):
        Um = torch.zeros_like(U)
        Um[:, i] = U[:, i]
        errs[f"rows[{i}]"] = _rel(rows[:, i], tm.vjp(Um, nv.HEAD_L, 0.7))
    bad = {k: v for k, v in errs.items() if not (v <= tol)}
    if bad:
        msgs = _localise(eng, tm, V, None, P, nv.HEAD_GGN, scale, [1, 2])
        pytest.fail(f"{name}: {bad}; first differing op: {msgs}")


def test_probe_chunking_and_single_vector():
    """P larger than the workspace chunk, and a plain (D,) vector, give the same answers."""
    net, Z, model_type, _ = _cases()["mlp_ragged"]
    state = create_state(net, 3, dtype=F64)
    big = LinearizedNet(state, Z, model_type, max_chunk=16)
    small = LinearizedNet(state, Z, model_type, max_chunk=3)
    V = torch.randn(7, big.D, generator=torch.Generator().manual_seed(5)).cuda()
    a = big.ggn_vp(V, 1.0, 0.1)
    b = small.ggn_vp(V, 1.0, 0.1)
    c = big.ggn_vp(V[2], 1.0, 0.1)
    torch.cuda.synchronize()
    assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    assert torch.allclose(a[2], c[0], rtol=1e-4, atol=1e-5)   # split-K atomics reorder the sum


@pytest.mark.parametrize("name", ["resnet1m", "mlp_ragged"])
def test_products_do_not_read_stale_workspace(name):
    """The cached bindings share ONE probe workspace (``ggn.shared_workspace``), so whatever another engine left there must
    not matter: every product is run with the workspace filled with NaN and with 1e30 beforehand and compared with the run
    on a zeroed workspace — a kernel that reads an element its own call has not written would return NaN / garbage."""
    if name == "resnet1m":
        net, model_type = ResNet1M(10), "classifier"
        Z = torch.rand(6, 32, 32, 3, generator=torch.Generator().manual_seed(2))
    else:
        net, Z, model_type, _ = _cases()[name]
    state = create_state(net, 4, dtype=F64)
    work = torch.empty(1 << 28, device="cuda", dtype=torch.float32)                 # 1 GiB shared pool
    eng = LinearizedNet(state, Z, model_type, work=work)
    other = LinearizedNet(create_state(SimpleClassifier(6, 2, 3), 1, dtype=F64), torch.randn(5, 2), "classifier", work=work)
    g = torch.Generator().manual_seed(9)
    for P in (1, 5, 17):
        V = torch.randn(P, eng.D, generator=g).cuda()
        U = torch.randn(P, eng.n, eng.K, generator=g).cuda()
        calls = (lambda: eng.ggn_vp(V, 3.0, 0.01), lambda: eng.jvp(V, "lt", 1.0), lambda: eng.vjp(U, "l", 1.0),
                 lambda: eng.vjp_rows(U[:2].contiguous(), "l", 1.0))
        ref = None
        for fill in (0.0, float("nan"), 1e30, "other engine"):
            outs = []
            for fn in calls:
                if fill == "other engine":
                    other.ggn_vp(torch.randn(4, other.D).cuda(), 1.0, 0.0)
                else:
                    work.fill_(fill)
                outs.append(fn().double())
            if ref is None:
                ref = outs
                continue
            for o, r in zip(outs, ref):
                assert bool(torch.isfinite(o).all()), f"{name} P={P}: workspace content {fill} reached the output"
                assert float((o - r).abs().max()) <= 2e-6 * float(r.abs().max()), (name, P, fill)     # float atomics reorder sums


def test_factor_rows_per_example_equal_one_hot_sweeps():
    """materialize_factor: K probes through the per-example sweep == d one-hot cotangents through the summed vjp."""
    from lip_amd.ggn import materialize_factor
    for name in ("xor_classifier", "resnet_small", "sine_regressor"):
        net, Z, model_type, _ = _cases()[name]
        eng = LinearizedNet(create_state(net, 4, dtype=F64, logvar=-0.2), Z, model_type)
        a = materialize_factor(eng, 0.9, per_example=True)
        b = materialize_factor(eng, 0.9, per_example=False)
        torch.cuda.synchronize()
        assert a.shape == b.shape == (eng.n * eng.K, eng.D)
        assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item(), name


def test_edge_sizes_single_example_single_probe_and_empty_block():
    """n = 1 example, P = 1 probe (split-K weight gradient path), and an empty probe block (a clean error, no launch)."""
    net = ResNet1M(5, input_shape=(8, 8, 3), widths=(16, 32), blocks_per_stage=1)
    st = create_state(net, 2, dtype=F64)
    Z = torch.rand(1, 8, 8, 3, dtype=F64, generator=torch.Generator().manual_seed(1))
    eng = LinearizedNet(st, Z, "classifier")
    flat, _ = flatten_nn_params(st.params)
    tm = TapeMachine(eng.cn, flat, build_consts(eng.cn, st.params, st.batch_stats, "cpu", F64), Z, chunk=1)
    tm.primal()
    v = torch.randn(1, eng.D, dtype=F64, generator=torch.Generator().manual_seed(2))
    y = eng.ggn_vp(v, 3.0, 0.2)
    torch.cuda.synchronize()
    assert _rel(y, tm.ggn_vp(v, 3.0, 0.2)) <= 2e-4
    rows = eng.vjp_rows(torch.ones(1, 1, eng.K), "l", 1.0)
    assert rows.shape == (1, 1, eng.D) and _rel(rows[:, 0], eng.vjp(torch.ones(1, 1, eng.K), "l", 1.0).double().cpu()) <= 1e-5
    with pytest.raises((nv.NativeError, ValueError, RuntimeError)):
        eng.ggn_vp(torch.zeros(0, eng.D), 1.0, 0.0)
    with pytest.raises(ValueError):
        LinearizedNet(st, Z[:0], "classifier")


def test_missing_netspec_fails_loudly():
    from lip_amd.utils import TrainState
    st = TrainState(params={"params": {"W": torch.ones(1, 1)}}, apply_fn=lambda p, x, **k: x)
    with pytest.raises(TypeError):
        LinearizedNet(st, torch.zeros(2, 1), "regressor")


def test_split_precision_mode_accuracy():
    """Opt-in bf16x3 MFMA path (lip_set_precision): stated tolerance 5e-5 * max|ref| against the float64 tape
    semantics (measured ~6e-6); the default f32 path on the same inputs stays below 2e-6."""
    from lip_amd.engine import get_precision, set_precision
    net, Z, model_type, P = _cases()["resnet_small"]
    state = create_state(net, 3, dtype=F64)
    eng = LinearizedNet(state, Z, model_type, workspace_bytes=1 << 28, max_chunk=P)
    flat, _ = flatten_nn_params(state.params)
    tm = TapeMachine(eng.cn, flat, build_consts(eng.cn, state.params, state.batch_stats, "cpu", F64), Z, chunk=eng.chunk)
    tm.primal()
    V = torch.randn(P, eng.D, dtype=F64, generator=torch.Generator().manual_seed(1))
    ref = tm.ggn_vp(V, 3.0, 0.2).clone()
    try:
        assert get_precision() == "f32"
        e32 = _rel(eng.ggn_vp(V, 3.0, 0.2), ref)
        set_precision("bf16x3")
        e16 = _rel(eng.ggn_vp(V, 3.0, 0.2), ref)
    finally:
        set_precision("f32")
    assert e32 < 2e-6 and e16 < 5e-5, (e32, e16)



def test_split_precision_at_the_bench_geometry():
    """bf16x3 mode on the binding bench.py runs (ResNet1M 32x32, n = 50, P = 32: the 128-row implicit-GEMM tiles and the
    probe-batched weight gradients are selected, as at P = 256).  Round 2's record showed 3e-4 and run-to-run different
    results there (the BN-scale cotangent reduced in the epilogue of the 64-column tile, see igemm_epilogue); the small
    test above never selects those kernels.  Stated tolerances: against the float64 oracle both modes stay inside the
    repo's fp32 bound 2e-4 * max|ref| (measured 6.2e-5 for exact f32 — float32 parameters and inputs at this depth —
    and 6.5e-5 for bf16x3); what the split operands ADD is bounded over all 32 probes by 2e-5 * max|Y| against the f32
    sweep on the same binding (measured 6.6e-6; 3e-4 before the fix); two split runs agree to atomics-reordering
    level, 1e-6 (measured 5e-8; 2.5e-4 before the fix)."""
    from lip_amd import krylov
    from lip_amd.engine import set_precision
    from oracle import ggn as og
    net = ResNet1M(10)
    st64 = create_state(net, 1231231234, dtype=F64)
    Z = torch.rand(50, 32, 32, 3, dtype=F64, generator=torch.Generator().manual_seed(280300))
    P, scale, alpha = 32, 980.0, 0.005
    eng = LinearizedNet(st64, Z, "classifier", workspace_bytes=4 << 30, max_chunk=P)
    assert eng.chunk == P
    V = krylov.fill_rademacher(P, eng.D, 1234, "cuda")
    ref_vp = og.compute_ggn_vp_batched(st64, Z, "classifier", full_set_size=49000)
    ref = torch.stack([ref_vp(v) + alpha * v for v in V[:2].double().cpu()])
    try:
        Y32 = eng.ggn_vp(V, scale, alpha).clone()
        set_precision("bf16x3")
        Ya = eng.ggn_vp(V, scale, alpha).clone()
        Yb = eng.ggn_vp(V, scale, alpha).clone()
    finally:
        set_precision("f32")
    torch.cuda.synchronize()
    m = Y32.abs().max().item()
    e32, e16 = _rel(Y32[:2], ref), _rel(Ya[:2], ref)
    rr = (Ya - Yb).abs().max().item() / m
    d = (Ya - Y32).abs().max().item() / m
    assert e32 < 2e-4 and e16 < 2e-4 and rr < 1e-6 and d < 2e-5, (e32, e16, rr, d)
