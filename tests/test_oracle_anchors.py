"""Anchors of the oracle (and of the HIP path) that go through neither ``oracle/``'s own output-Hessian code nor
``NetSpec.forward`` nor the product's flat-theta layout code:

(a) a LINEAR softmax classifier — the GGN equals the Hessian of the summed cross-entropy (the network is linear in
    theta), exactly the idea of the reference's ``tests/test_ggn.py:21-54`` (``jax.hessian`` of the NLL for the linear
    regressor) carried over to the classifier; the loss, the model and the flat order are written out by hand here
    and differentiated by ``torch.autograd.functional.hessian``;
(b) a small conv + BN(eval) + ReLU + residual + stride-2 projection + mean-pool + Dense net written directly from the
    reference's Flax definition (``src/scalemodels.py:70-157``) with plain tensor slicing — SAME padding, HWIO kernels,
    NHWC activations, sorted-key flat order all restated by hand — per-example ``jacrev`` and the explicit
    ``diag(p) - p p^T``;
(c) the square-root forms of ``src/ggn.py:16-39``: ``L L^T = diag(p) - p p^T``, ``L sqrt(p) = 0``, ``W^T`` is the
    transpose of ``W``.

Every test runs on the float64 oracle (CPU) and, marked ``gpu``, on the HIP path; both are compared with the
hand-written anchor, never with each other.
"""
import math

import pytest
import torch
from torch.func import jacrev

from impl import cpu64, impl  # noqa: F401
from lip_amd.netspec import NetSpec
from lip_amd.scalemodels import ResNet1M
from lip_amd.utils import TrainState, flatten_nn_params

F64 = torch.float64


def _randn(seed, *shape):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed), dtype=F64)


# ------------------------------------------------------------------------------------------------ (a)
def _linear_softmax(in_f=3, K=4, n=6):
    """theta = [bias (K) | kernel (in, K) row-major]: Flax ``Dense`` names its leaves 'bias' and 'kernel' and
    ``ravel_pytree`` walks dict keys in sorted order."""
    theta = 0.7 * _randn(11, K + in_f * K)
    X = _randn(12, n, in_f)
    y = torch.randint(0, K, (n,), generator=torch.Generator().manual_seed(13))
    params = {"params": {"Dense_0": {"bias": theta[:K].clone(), "kernel": theta[K:].reshape(in_f, K).clone()}}}

    def apply_fn(variables, x, train=False, mutable=False, **_):          # hand-written; not NetSpec.forward
        p = variables["params"]["Dense_0"]
        return x @ p["kernel"] + p["bias"]

    net = NetSpec((in_f,))
    net.dense(0, "Dense_0", K)
    net.model_type = "classifier"
    st = TrainState(params=params, apply_fn=apply_fn, batch_stats={}, net=net)

    def summed_ce(th):                                                   # hand-written loss on the hand-written layout
        logits = X @ th[K:].reshape(in_f, K) + th[:K]
        return -(torch.log_softmax(logits, -1)[torch.arange(n), y]).sum()

    return st, X, theta, summed_ce


def test_linear_softmax_ggn_is_hessian_of_ce(impl):
    st, X, theta, summed_ce = _linear_softmax()
    H = torch.autograd.functional.hessian(summed_ce, theta)
    GGN, flat, _ = impl.ggn.compute_ggn_dense(impl.state(st), impl.tensor(X), "classifier")
    assert torch.equal(cpu64(flat), theta if not impl.is_hip else theta.float().double())   # the flat order itself
    tol = impl.tol(1e-12, 2e-6) * H.abs().max().item()
    assert (cpu64(GGN) - H).abs().max().item() <= tol
    # and with the N/M recalibration of src/ggn.py:109-111
    GGN3, _, _ = impl.ggn.compute_ggn_dense(impl.state(st), impl.tensor(X), "classifier", full_set_size=3 * X.shape[0])
    assert (cpu64(GGN3) - 3.0 * H).abs().max().item() <= 3 * tol


def test_linear_softmax_ggn_vp_and_factors(impl):
    st, X, theta, summed_ce = _linear_softmax()
    H = torch.autograd.functional.hessian(summed_ce, theta)
    D = theta.numel()
    V = _randn(14, 5, D)
    vp = impl.ggn.compute_ggn_vp(impl.state(st), impl.tensor(X), "classifier")
    Y = impl.rows(vp, impl.tensor(V))
    tol = impl.tol(1e-12, 2e-6) * (V @ H).abs().max().item()
    assert (cpu64(Y) - V @ H).abs().max().item() <= tol
    Wfun, WTfun = impl.ggn.compute_W_vps(impl.state(st), impl.tensor(X), "classifier")
    U = impl.rows(WTfun, impl.tensor(V))                                  # (5, n, K)
    WWt = impl.rows(Wfun, U)                                              # W W^T v = GGN v
    assert (cpu64(WWt) - V @ H).abs().max().item() <= 2 * tol


# ------------------------------------------------------------------------------------------------ (b)
def _conv_same(x, w, stride):
    """XLA / Flax 'SAME' convolution written from its definition.  x (H, W, Cin), w (kh, kw, Cin, Cout) HWIO.
    out = ceil(in / stride); total padding max((out-1) stride + k - in, 0), the odd unit on the HIGH side."""
    H, Wd, _ = x.shape
    kh, kw, _, co = w.shape
    oh, ow = -(-H // stride), -(-Wd // stride)
    th, tw = max((oh - 1) * stride + kh - H, 0), max((ow - 1) * stride + kw - Wd, 0)
    xp = torch.nn.functional.pad(x, (0, 0, tw // 2, tw - tw // 2, th // 2, th - th // 2))
    out = 0.0
    for i in range(kh):
        for j in range(kw):
            patch = xp[i:i + (oh - 1) * stride + 1:stride, j:j + (ow - 1) * stride + 1:stride]     # (oh, ow, Cin)
            out = out + patch @ w[i, j]
    return out


def _bn_eval(x, scale, bias, mean, var):
    return (x - mean) / torch.sqrt(var + 1e-5) * scale + bias           # flax.linen.BatchNorm: epsilon = 1e-5


# Flat order of the mini ResNet1M below = sorted keys at every level (ravel_pytree), C-order leaves.
def _mini_layout(c0, c1, K):
    def block(ch_in, ch, proj):
        names = [("BatchNorm_0", "bias", (ch,)), ("BatchNorm_0", "scale", (ch,)),
                 ("BatchNorm_1", "bias", (ch,)), ("BatchNorm_1", "scale", (ch,))]
        if proj:
            names += [("BatchNorm_2", "bias", (ch,)), ("BatchNorm_2", "scale", (ch,))]
        names += [("Conv_0", "kernel", (3, 3, ch_in, ch)), ("Conv_1", "kernel", (3, 3, ch, ch))]
        if proj:
            names += [("Conv_2", "kernel", (1, 1, ch_in, ch))]
        return names
    lay = [(("BasicBlock_0",) + nm[:2], nm[2]) for nm in block(c0, c0, False)]
    lay += [(("BasicBlock_1",) + nm[:2], nm[2]) for nm in block(c0, c1, True)]
    lay += [(("BatchNorm_0", "bias"), (c0,)), (("BatchNorm_0", "scale"), (c0,)),
            (("Conv_0", "kernel"), (3, 3, 3, c0)),
            (("Dense_0", "bias"), (K,)), (("Dense_0", "kernel"), (c1, K))]
    return lay


def _unflat(theta, layout):
    tree, off = {}, 0
    for path, shape in layout:
        n = math.prod(shape)
        t = tree
        for k in path[:-1]:
            t = t.setdefault(k, {})
        t[path[-1]] = theta[off:off + n].reshape(shape)
        off += n
    assert off == theta.numel()
    return tree


def _mini_resnet(c0, c1, K, hw):
    """ResNet1M(K, input (hw, hw, 3), widths (c0, c1), one block per stage): stem conv+BN+ReLU, an identity-shortcut
    block, a stride-2 block with a 1x1 stride-2 projection shortcut, global mean pool, Dense — the structure of
    ``src/scalemodels.py:115-157`` at test size."""
    layout = _mini_layout(c0, c1, K)
    D = sum(math.prod(s) for _, s in layout)
    theta = _randn(21, D)
    tree = _unflat(theta, layout)
    # sensible magnitudes: kernels ~ 1/sqrt(fan_in), BN scale ~ 1
    for path, shape in layout:
        t = tree
        for k in path[:-1]:
            t = t[k]
        if path[-1] == "kernel":
            t[path[-1]] = t[path[-1]] / math.sqrt(math.prod(shape[:-1]))
        elif path[-1] == "scale":
            t[path[-1]] = 1.0 + 0.2 * t[path[-1]]
        else:
            t[path[-1]] = 0.1 * t[path[-1]]
    theta = torch.cat([_get(tree, p).reshape(-1) for p, _ in layout])
    stats = {}
    g = torch.Generator().manual_seed(22)
    for path, shape in layout:
        if path[-2].startswith("BatchNorm") and path[-1] == "bias":
            t = stats
            for k in path[:-2]:
                t = t.setdefault(k, {})
            t[path[-2]] = {"mean": 0.1 * torch.randn(shape, generator=g, dtype=F64),
                           "var": 0.5 + torch.rand(shape, generator=g, dtype=F64)}

    def forward(th, x):                                                   # one example x (hw, hw, 3) -> logits (K,)
        p = _unflat(th, layout)
        bn = lambda z, pp, ss: _bn_eval(z, pp["scale"], pp["bias"], ss["mean"], ss["var"])
        h = torch.relu(bn(_conv_same(x, p["Conv_0"]["kernel"], 1), p["BatchNorm_0"], stats["BatchNorm_0"]))
        b, s = p["BasicBlock_0"], stats["BasicBlock_0"]
        r = h
        h = torch.relu(bn(_conv_same(h, b["Conv_0"]["kernel"], 1), b["BatchNorm_0"], s["BatchNorm_0"]))
        h = bn(_conv_same(h, b["Conv_1"]["kernel"], 1), b["BatchNorm_1"], s["BatchNorm_1"])
        h = torch.relu(h + r)
        b, s = p["BasicBlock_1"], stats["BasicBlock_1"]
        r = bn(_conv_same(h, b["Conv_2"]["kernel"], 2), b["BatchNorm_2"], s["BatchNorm_2"])
        h = torch.relu(bn(_conv_same(h, b["Conv_0"]["kernel"], 2), b["BatchNorm_0"], s["BatchNorm_0"]))
        h = bn(_conv_same(h, b["Conv_1"]["kernel"], 1), b["BatchNorm_1"], s["BatchNorm_1"])
        h = torch.relu(h + r)
        return h.mean(dim=(0, 1)) @ p["Dense_0"]["kernel"] + p["Dense_0"]["bias"]

    def apply_fn(variables, x, train=False, mutable=False, **_):          # hand-written; not NetSpec.forward
        th = torch.cat([_get(variables["params"], pth).reshape(-1) for pth, _ in layout])
        if x.dim() == 3:
            return forward(th, x)
        return torch.stack([forward(th, xi) for xi in x])

    net = ResNet1M(K, input_shape=(hw, hw, 3), widths=(c0, c1), blocks_per_stage=1)
    st = TrainState(params={"params": _unflat(theta, layout)}, apply_fn=apply_fn, batch_stats=stats, net=net)
    return st, theta, forward


def _get(tree, path):
    for k in path:
        tree = tree[k]
    return tree


def _explicit_ggn_rows(forward, theta, Z, V, scale):
    """sum_i J_i^T (diag p_i - p_i p_i^T) J_i v with J_i from jacrev of the hand-written net."""
    out = torch.zeros_like(V)
    factors = []
    for z in Z:
        J = jacrev(lambda th: forward(th, z))(theta)                    # (K, D)
        p = torch.softmax(forward(theta, z), -1)
        Hl = torch.diag(p) - torch.outer(p, p)
        out += V @ J.T @ Hl @ J
        factors.append((J, p))
    return scale * out, factors


@pytest.mark.parametrize("widths", [(4, 8), (16, 32)])
def test_conv_bn_residual_net_against_explicit_jacobians(impl, widths):
    """(4, 8) exercises the generic kernels, (16, 32) the straight-line MFMA kernels (C % 16 == 0) and the
    parity-class stride-2 data gradient."""
    c0, c1 = widths
    K, hw, n = 3, 8, 3
    st, theta, forward = _mini_resnet(c0, c1, K, hw)
    flat, _ = flatten_nn_params(st.params)
    assert torch.equal(flat, theta)                                       # product layout == hand-written sorted-key layout
    Z = torch.rand(n, hw, hw, 3, generator=torch.Generator().manual_seed(23), dtype=F64)
    # the product's functional forward (used by the CPU baseline and by every other oracle test) against the hand-written net
    assert torch.allclose(st.net.forward(st.params, st.batch_stats, Z), torch.stack([forward(theta, z) for z in Z]),
                          rtol=1e-11, atol=1e-12)
    V = _randn(24, 2, theta.numel())
    N = 5 * n
    ref, factors = _explicit_ggn_rows(forward, theta, Z, V, N / n)
    vp = impl.ggn.compute_ggn_vp(impl.state(st), impl.tensor(Z), "classifier", full_set_size=N)
    Y = impl.rows(vp, impl.tensor(V))
    tol = impl.tol(1e-11, 3e-5) * ref.abs().max().item()
    assert (cpu64(Y) - ref).abs().max().item() <= tol
    # square-root factor: W^T v = sqrt(N/M) [L_i^T J_i v]_i with L = diag(sqrt p) - p sqrt(p)^T  (src/ggn.py:23-39)
    Wfun, WTfun = impl.ggn.compute_W_vps(impl.state(st), impl.tensor(Z), "classifier", full_set_size=N)
    U = cpu64(impl.rows(WTfun, impl.tensor(V)))                           # (2, n, K)
    for i, (J, p) in enumerate(factors):
        L = torch.diag(torch.sqrt(p)) - torch.outer(p, torch.sqrt(p))
        want = math.sqrt(N / n) * (V @ J.T @ L)                           # rows: L^T J v
        assert (U[:, i] - want).abs().max().item() <= impl.tol(1e-11, 3e-5) * max(1.0, want.abs().max().item())
    WU = cpu64(impl.rows(Wfun, impl.tensor(U)))
    assert (WU - ref).abs().max().item() <= 2 * tol


# ------------------------------------------------------------------------------------------------ (c)
def test_sqrt_factor_forms(impl):
    """With a zero input the Dense layer's Jacobian is [I_K | 0] (bias first in the flat order), so W and W^T expose
    L and L^T themselves: L L^T = diag(p) - p p^T, L sqrt(p) = 0, (L^T)^T = L  (src/ggn.py:23-27,35-39)."""
    K, in_f = 5, 2
    bias = _randn(31, K)
    kernel = _randn(32, in_f, K)
    params = {"params": {"Dense_0": {"bias": bias, "kernel": kernel}}}

    def apply_fn(variables, x, train=False, mutable=False, **_):
        p = variables["params"]["Dense_0"]
        return x @ p["kernel"] + p["bias"]

    net = NetSpec((in_f,))
    net.dense(0, "Dense_0", K)
    net.model_type = "classifier"
    st = TrainState(params=params, apply_fn=apply_fn, batch_stats={}, net=net)
    Z = torch.zeros(1, in_f, dtype=F64)
    D = K + in_f * K
    p = torch.softmax(bias, -1)
    Wfun, WTfun = impl.ggn.compute_W_vps(impl.state(st), impl.tensor(Z), "classifier")
    E_K = torch.eye(K, dtype=F64).reshape(K, 1, K)
    L = cpu64(impl.rows(Wfun, impl.tensor(E_K)))[:, :K].T                 # column k = L e_k
    assert cpu64(impl.rows(Wfun, impl.tensor(E_K)))[:, K:].abs().max().item() == 0.0
    E_D = torch.eye(D, dtype=F64)[:K]
    LT = cpu64(impl.rows(WTfun, impl.tensor(E_D))).reshape(K, K).T        # column k = L^T e_k
    tol = impl.tol(1e-14, 1e-6)
    assert (L @ L.T - (torch.diag(p) - torch.outer(p, p))).abs().max().item() <= tol
    assert (L @ torch.sqrt(p)).abs().max().item() <= tol
    assert (LT - L.T).abs().max().item() <= tol
    assert (L - (torch.diag(torch.sqrt(p)) - torch.outer(p, torch.sqrt(p)))).abs().max().item() <= tol


# ------------------------------------------------------------------------------------------------ (d) GELU MLP
def test_gelu_mlp_regressor_against_explicit_jacobians(impl):
    """``SimpleRegressor`` (``src/toymodels.py:4-24``: Dense -> GELU -> Dense -> GELU -> Dense(1), scalar logvar outside
    theta) written by hand — flax.linen.gelu's default tanh form, bias-before-kernel sorted-key layout — and
    differentiated by ``jacrev``: GGN v = exp(-logvar) (N / M) sum_i J_i^T J_i v  (``src/ggn.py:111-113``)."""
    from lip_amd.toymodels import SimpleRegressor
    h, n, N = 5, 4, 12
    shapes = [(("Dense_0", "bias"), (h,)), (("Dense_0", "kernel"), (1, h)), (("Dense_1", "bias"), (h,)),
              (("Dense_1", "kernel"), (h, h)), (("Dense_2", "bias"), (1,)), (("Dense_2", "kernel"), (h, 1))]
    D = sum(math.prod(s) for _, s in shapes)
    theta = 0.8 * _randn(41, D)
    logvar = torch.tensor(-0.3, dtype=F64)

    def gelu(x):                                                          # 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
        return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))

    def forward(th, x):                                                   # x (1,) -> (1,)
        p = _unflat(th, shapes)
        a = gelu(x @ p["Dense_0"]["kernel"] + p["Dense_0"]["bias"])
        a = gelu(a @ p["Dense_1"]["kernel"] + p["Dense_1"]["bias"])
        return a @ p["Dense_2"]["kernel"] + p["Dense_2"]["bias"]

    def apply_fn(variables, x, return_logvar=False, **_):                 # hand-written; not NetSpec.forward
        th = torch.cat([_get(variables["params"], pth).reshape(-1) for pth, _ in shapes])
        return forward(th, x) if x.dim() == 1 else torch.stack([forward(th, xi) for xi in x])

    net = SimpleRegressor(h, 2)
    params = {"params": _unflat(theta, shapes), "logvar": {"logvar": logvar}}
    st = TrainState(params=params, apply_fn=apply_fn, batch_stats={}, net=net)
    flat, _ = flatten_nn_params(st.params)
    assert torch.equal(flat, theta)                                       # logvar is not part of theta
    Z = _randn(42, n, 1)
    assert torch.allclose(net.forward(st.params, st.batch_stats, Z), torch.stack([forward(theta, z) for z in Z]),
                          rtol=1e-12, atol=1e-13)
    V = _randn(43, 3, D)
    ref = torch.zeros_like(V)
    for z in Z:
        J = jacrev(lambda th: forward(th, z))(theta)                      # (1, D)
        ref += V @ J.T @ J
    ref = ref * torch.exp(-logvar) * (N / n)
    vp = impl.ggn.compute_ggn_vp(impl.state(st), impl.tensor(Z), "regressor", full_set_size=N)
    Y = impl.rows(vp, impl.tensor(V))
    assert (cpu64(Y) - ref).abs().max().item() <= impl.tol(1e-11, 3e-5) * ref.abs().max().item()


# ------------------------------------------------------------------------------------------------ (e) max pool + bottleneck
def _conv_pad(x, w, stride, pad):
    """convolution with symmetric zero padding ``pad`` (torch convention), x (H, W, Cin), w HWIO — from the definition"""
    H, Wd, _ = x.shape
    kh, kw, _, co = w.shape
    oh, ow = (H + 2 * pad - kh) // stride + 1, (Wd + 2 * pad - kw) // stride + 1
    xp = torch.nn.functional.pad(x, (0, 0, pad, pad, pad, pad))
    out = 0.0
    for i in range(kh):
        for j in range(kw):
            out = out + xp[i:i + (oh - 1) * stride + 1:stride, j:j + (ow - 1) * stride + 1:stride] @ w[i, j]
    return out


def _maxpool_pad(x, k, stride, pad):
    """k x k / stride max pool, padding ``pad`` with -inf (a padded position never wins)"""
    H, Wd, C = x.shape
    oh, ow = (H + 2 * pad - k) // stride + 1, (Wd + 2 * pad - k) // stride + 1
    xp = torch.nn.functional.pad(x, (0, 0, pad, pad, pad, pad), value=float("-inf"))
    out = torch.full((oh, ow, C), float("-inf"), dtype=x.dtype)
    for i in range(k):
        for j in range(k):
            out = torch.maximum(out, xp[i:i + (oh - 1) * stride + 1:stride, j:j + (ow - 1) * stride + 1:stride])
    return out


def test_maxpool_bottleneck_net_against_explicit_jacobians(impl):
    """The ResNet-50 pattern of BASELINE configs[4] at test size (``scalemodels.ResNet50(K, (20, 20, 3), stem 8, widths
    (4, 8), blocks (1, 1))``): 7x7/2 stem + BN + ReLU, 3x3/2 max pool with padding, a bottleneck with projection
    shortcut, a stride-2 bottleneck, mean pool, Dense — written by hand (symmetric padding, -inf pool padding,
    sorted-key flat layout) with per-example ``jacrev`` and the explicit diag(p) - p p^T."""
    from lip_amd.scalemodels import ResNet50
    K, hw, stem, n = 5, 20, 8, 2
    def bott(cin, planes, proj):
        names = [("BatchNorm_0", "bias", (planes,)), ("BatchNorm_0", "scale", (planes,)),
                 ("BatchNorm_1", "bias", (planes,)), ("BatchNorm_1", "scale", (planes,)),
                 ("BatchNorm_2", "bias", (4 * planes,)), ("BatchNorm_2", "scale", (4 * planes,))]
        if proj:
            names += [("BatchNorm_3", "bias", (4 * planes,)), ("BatchNorm_3", "scale", (4 * planes,))]
        names += [("Conv_0", "kernel", (1, 1, cin, planes)), ("Conv_1", "kernel", (3, 3, planes, planes)),
                  ("Conv_2", "kernel", (1, 1, planes, 4 * planes))]
        if proj:
            names += [("Conv_3", "kernel", (1, 1, cin, 4 * planes))]
        return names
    layout = [(("BatchNorm_0", "bias"), (stem,)), (("BatchNorm_0", "scale"), (stem,))]
    layout += [(("Bottleneck_0",) + nm[:2], nm[2]) for nm in bott(stem, 4, True)]
    layout += [(("Bottleneck_1",) + nm[:2], nm[2]) for nm in bott(16, 8, True)]
    layout += [(("Conv_0", "kernel"), (7, 7, 3, stem)), (("Dense_0", "bias"), (K,)), (("Dense_0", "kernel"), (32, K))]
    D = sum(math.prod(s) for _, s in layout)
    theta = _randn(51, D)
    tree = _unflat(theta, layout)
    for path, shape in layout:
        t = _get(tree, path[:-1])
        if path[-1] == "kernel":
            t[path[-1]] = t[path[-1]] / math.sqrt(math.prod(shape[:-1]))
        elif path[-1] == "scale":
            t[path[-1]] = 1.0 + 0.2 * t[path[-1]]
        else:
            t[path[-1]] = 0.1 * t[path[-1]]
    theta = torch.cat([_get(tree, p).reshape(-1) for p, _ in layout])
    stats, g = {}, torch.Generator().manual_seed(52)
    for path, shape in layout:
        if path[-2].startswith("BatchNorm") and path[-1] == "bias":
            t = stats
            for k in path[:-2]:
                t = t.setdefault(k, {})
            t[path[-2]] = {"mean": 0.1 * torch.randn(shape, generator=g, dtype=F64), "var": 0.5 + torch.rand(shape, generator=g, dtype=F64)}

    def forward(th, x):
        p = _unflat(th, layout)
        bn = lambda z, pp, ss: _bn_eval(z, pp["scale"], pp["bias"], ss["mean"], ss["var"])
        h = torch.relu(bn(_conv_pad(x, p["Conv_0"]["kernel"], 2, 3), p["BatchNorm_0"], stats["BatchNorm_0"]))
        h = _maxpool_pad(h, 3, 2, 1)
        for name, stride in (("Bottleneck_0", 1), ("Bottleneck_1", 2)):
            b, s = p[name], stats[name]
            r = bn(_conv_pad(h, b["Conv_3"]["kernel"], stride, 0), b["BatchNorm_3"], s["BatchNorm_3"])
            y = torch.relu(bn(_conv_pad(h, b["Conv_0"]["kernel"], 1, 0), b["BatchNorm_0"], s["BatchNorm_0"]))
            y = torch.relu(bn(_conv_pad(y, b["Conv_1"]["kernel"], stride, 1), b["BatchNorm_1"], s["BatchNorm_1"]))
            h = torch.relu(bn(_conv_pad(y, b["Conv_2"]["kernel"], 1, 0), b["BatchNorm_2"], s["BatchNorm_2"]) + r)
        return h.mean(dim=(0, 1)) @ p["Dense_0"]["kernel"] + p["Dense_0"]["bias"]

    def apply_fn(variables, x, train=False, mutable=False, **_):
        th = torch.cat([_get(variables["params"], pth).reshape(-1) for pth, _ in layout])
        return forward(th, x) if x.dim() == 3 else torch.stack([forward(th, xi) for xi in x])

    net = ResNet50(K, input_shape=(hw, hw, 3), stem=stem, widths=(4, 8), blocks=(1, 1))
    st = TrainState(params={"params": _unflat(theta, layout)}, apply_fn=apply_fn, batch_stats=stats, net=net)
    flat, _ = flatten_nn_params(st.params)
    assert torch.equal(flat, theta)
    Z = torch.rand(n, hw, hw, 3, generator=torch.Generator().manual_seed(53), dtype=F64)
    assert torch.allclose(net.forward(st.params, st.batch_stats, Z), torch.stack([forward(theta, z) for z in Z]), rtol=1e-11, atol=1e-12)
    V = _randn(54, 2, D)
    ref, _ = _explicit_ggn_rows(forward, theta, Z, V, 7 / n)
    vp = impl.ggn.compute_ggn_vp(impl.state(st), impl.tensor(Z), "classifier", full_set_size=7)
    Y = impl.rows(vp, impl.tensor(V))
    assert (cpu64(Y) - ref).abs().max().item() <= impl.tol(1e-11, 3e-5) * ref.abs().max().item()
