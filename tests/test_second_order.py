"""The second-order pass of the inducing-point gradient (``lip_amd/second_order.py``: reverse mode over the tangent
tape, every convolution an ``LIP_OP_IGEMM``) against ``torch.func`` differentiation of the same pairing
phi(Z) = sum_{i,k} < J(z_i) m_ik, c L(z_i) e_k >  (reference: ``jax.value_and_grad`` through
``src/train_inducing.py:87-173``).  On the CPU the ops run on the float64 tape emulator — this validates the op
sequence, operand offsets and the adjoint algebra; ``-m gpu`` runs the same ops on the HIP kernels."""
import math

import pytest
import torch
from torch.func import grad, jvp

from lip_amd import second_order as so
from lip_amd.engine import build_consts, compile_net
from lip_amd.scalemodels import LargeClassifier, ResNet1M
from lip_amd.toymodels import SimpleClassifier, SimpleRegressor, create_state
from lip_amd.utils import flatten_nn_params
from tape_emulator import TapeMachine

F64 = torch.float64


def _cases():
    g = torch.Generator().manual_seed(3)
    return {
        "xor_tanh_mlp": (SimpleClassifier(8, 2, 2), torch.randn(4, 2, dtype=F64, generator=g), "classifier"),
        "sine_gelu_mlp": (SimpleRegressor(6, 2), torch.randn(3, 1, dtype=F64, generator=g), "regressor"),
        "resnet_bn_res_stride2": (ResNet1M(3, input_shape=(6, 6, 3), widths=(4, 8), blocks_per_stage=1),
                                  torch.rand(3, 6, 6, 3, dtype=F64, generator=g), "classifier"),
        "resnet_gray_tiled": (ResNet1M(3, input_shape=(6, 6, 1), widths=(4, 4), blocks_per_stage=1),
                              torch.rand(2, 6, 6, 1, dtype=F64, generator=g), "classifier"),
        "flatten_mlp": (LargeClassifier((3, 3, 1), [5, 4], 2, 3), torch.rand(3, 3, 3, 1, dtype=F64, generator=g), "classifier"),
        "lenet_avgpool_flatdense": (_mini_lenet(), torch.rand(2, 12, 12, 1, dtype=F64, generator=g), "classifier"),
        "stem_maxpool": (_mini_stem(), torch.rand(2, 9, 9, 3, dtype=F64, generator=g), "classifier"),
    }


def _mini_lenet():
    """LeNet5's structure (src/scalemodels.py:11-49) at test size: conv + bias + ReLU, 2x2 average pool, Dense on the
    flattened map, Dense."""
    from lip_amd.netspec import NetSpec
    net = NetSpec((12, 12, 1))
    x = net.conv(0, "Conv_0", 3, 5, 1, padding=2, act="relu", use_bias=True)
    x = net.avgpool(x, 2, 2)
    x = net.conv(x, "Conv_1", 4, 3, 1, padding="VALID", act="relu", use_bias=True)
    x = net.avgpool(x, 2, 2)
    x = net.dense(x, "Dense_0", 6, act="relu")
    net.dense(x, "Dense_1", 3)
    net.model_type = "classifier"
    return net


def _mini_stem():
    """the ResNet-50 stem's pattern: strided conv + BN + ReLU, overlapping 3x3 / 2 max pool with padding, then a
    conv, mean pool and Dense."""
    from lip_amd.netspec import NetSpec
    net = NetSpec((9, 9, 3))
    x = net.conv(0, "Conv_0", 4, 3, 2, padding=1, bn="BatchNorm_0", act="relu")
    x = net.maxpool(x, 3, 2, padding=1)
    x = net.conv(x, "Conv_1", 4, 3, 1, padding=1, bn="BatchNorm_1", act="relu")
    x = net.meanpool(x)
    net.dense(x, "Dense_0", 3)
    net.model_type = "classifier"
    return net


def _reference(st, Z, Mdir, c, model_type):
    """grad_Z of the pairing by forward-over-reverse torch.func on the functional forward, float64."""
    net = st.net
    flat, unravel = flatten_nn_params(st.params)

    def f(theta, z):
        return net.forward(unravel(theta), st.batch_stats, z).reshape(-1)

    def phi_one(z, Mj):
        JM = torch.stack([jvp(lambda th: f(th, z), (flat,), (m,))[1] for m in Mj])        # row k = J(z) m_k
        if model_type == "classifier":
            p = torch.softmax(f(flat, z), dim=-1)
            sq = torch.sqrt(p)
            L = torch.diag(sq) - torch.outer(p, sq)
            return c * (JM * L.T).sum()
        return c * torch.diagonal(JM).sum()

    return torch.stack([grad(lambda z: phi_one(z, Mdir[j]))(Z[j]) for j in range(Z.shape[0])])


class _EmulatorExecutor:
    def __init__(self, cn, st, Z, K):
        flat, _ = flatten_nn_params(st.params)
        self.tm = TapeMachine(cn, flat, build_consts(cn, st.params, st.batch_stats, "cpu", F64), Z, chunk=K)
        self.tm.primal()
        self.cn, self.device, self.dtype = cn, torch.device("cpu"), F64
        self.prim, self.consts, self.theta = self.tm.prim, self.tm.consts, self.tm.theta
        self.max_probes = K

    def run(self, op, P, V, Y):
        self.tm.V, self.tm.Y = V.reshape(-1), Y
        self.tm.run_op(op, P)


@pytest.mark.parametrize("name", list(_cases()))
def test_second_order_pass_on_the_tape_emulator(name):
    net, Z, mt = _cases()[name]
    st = create_state(net, 5, dtype=F64, logvar=-0.4)
    n = Z.shape[0]
    cn = compile_net(net, n, st.params)
    Mdir = torch.randn(n, cn.K, cn.D, dtype=F64, generator=torch.Generator().manual_seed(9))
    c = math.exp(0.2) if mt == "regressor" else 1.0
    ex = _EmulatorExecutor(cn, st, Z, cn.K)
    got = so.input_grad_of_pairing(ex, Mdir, c, mt)
    ref = _reference(st, Z, Mdir, c, mt)
    assert got.shape == Z.shape
    assert (got - ref).abs().max().item() <= 1e-10 * max(1.0, ref.abs().max().item()), (got - ref).abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(_cases()) + ["resnet_fast_kernels"])
def test_second_order_pass_on_the_hip_engine(name):
    from lip_amd.engine import LinearizedNet
    if name == "resnet_fast_kernels":          # channel counts that take the straight-line MFMA kernels for the shared-weight ops
        net, Z, mt = ResNet1M(4, input_shape=(8, 8, 3), widths=(16, 32), blocks_per_stage=1), \
            torch.rand(3, 8, 8, 3, dtype=F64, generator=torch.Generator().manual_seed(4)), "classifier"
    else:
        net, Z, mt = _cases()[name]
    st = create_state(net, 5, dtype=F64, logvar=-0.4)
    n = Z.shape[0]
    eng = LinearizedNet(st.to(device="cuda", dtype=torch.float32), Z.cuda().float(), mt, workspace_bytes=1 << 28, max_chunk=16)
    Mdir = torch.randn(n, eng.K, eng.D, dtype=F64, generator=torch.Generator().manual_seed(9))
    c = math.exp(0.2) if mt == "regressor" else 1.0
    got = so.input_grad_of_pairing(so.EngineExecutor(eng), Mdir.cuda().float(), c, mt).double().cpu()
    ref = _reference(st, Z, Mdir, c, mt)
    assert (got - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item()), (got - ref).abs().max().item()


def _reference_shared(st, Z, Udir, Xw, c, model_type):
    """grad_Z of  sum_{i,t} < J(z_i) u_t , c L(z_i) x_it >  (shared directions, general output weights), float64."""
    net = st.net
    flat, unravel = flatten_nn_params(st.params)

    def f(theta, z):
        return net.forward(unravel(theta), st.batch_stats, z).reshape(-1)

    def phi_one(z, Xi):
        JM = torch.stack([jvp(lambda th: f(th, z), (flat,), (u,))[1] for u in Udir])      # row t = J(z) u_t
        if model_type == "classifier":
            p = torch.softmax(f(flat, z), dim=-1)
            sq = torch.sqrt(p)
            L = torch.diag(sq) - torch.outer(p, sq)
            return c * (JM * (Xi @ L.T)).sum()
        return c * (JM * Xi).sum()

    return torch.stack([grad(lambda z: phi_one(z, Xw[j]))(Z[j]) for j in range(Z.shape[0])])


@pytest.mark.parametrize("name", ["xor_tanh_mlp", "sine_gelu_mlp", "resnet_bn_res_stride2", "stem_maxpool"])
def test_shared_direction_pairing_on_the_tape_emulator(name):
    """The cotangent of the stochastic objective is a sum of rank-one terms u_t x_t^T (``stochastic_grad.py``): T
    directions shared by the examples, arbitrary output weights, consumed in chunks."""
    net, Z, mt = _cases()[name]
    st = create_state(net, 5, dtype=F64, logvar=-0.4)
    n = Z.shape[0]
    cn = compile_net(net, n, st.params)
    T = 5
    g = torch.Generator().manual_seed(9)
    Udir = torch.randn(T, cn.D, dtype=F64, generator=g)
    Xw = torch.randn(n, T, cn.K, dtype=F64, generator=g)
    c = math.exp(0.2) if mt == "regressor" else 1.0
    ex = _EmulatorExecutor(cn, st, Z, T)
    got = so.input_grad_of_pairing(ex, Udir, c, mt, out_weights=Xw, shared=True)
    ref = _reference_shared(st, Z, Udir, Xw, c, mt)
    assert (got - ref).abs().max().item() <= 1e-10 * max(1.0, ref.abs().max().item()), (got - ref).abs().max().item()
    # the chunked driver over (U, X) blocks gives the same sum
    ex2 = _EmulatorExecutor(cn, st, Z, 2)
    X2 = Xw.permute(1, 0, 2).reshape(T, n * cn.K)
    got2 = so.input_grad_of_rank_one_terms(ex2, [(Udir[:3], X2[:3]), (Udir[3:], X2[3:])], c, mt, max_directions=2)
    assert (got2 - ref).abs().max().item() <= 1e-10 * max(1.0, ref.abs().max().item())


def test_zero_bn_scale_gives_a_finite_gradient():
    """A BatchNorm scale that is exactly zero (zero-initialised residual gamma): the x-hat adjoint is dgamma rstd DY,
    not (dgamma / gamma) (gamma rstd) DY = inf * 0."""
    net, Z, mt = _cases()["resnet_bn_res_stride2"]
    st = create_state(net, 5, dtype=F64, logvar=-0.4)
    bn_path = next(u.bn_scale for u in net.units if getattr(u, "bn_scale", None) is not None)
    node = st.params                                  # the path starts at the top level ('params', ..., 'scale')
    for key in bn_path[:-1]:
        node = node[key]
    node[bn_path[-1]].zero_()
    n = Z.shape[0]
    cn = compile_net(net, n, st.params)
    Mdir = torch.randn(n, cn.K, cn.D, dtype=F64, generator=torch.Generator().manual_seed(9))
    got = so.input_grad_of_pairing(_EmulatorExecutor(cn, st, Z, cn.K), Mdir, 1.0, mt)
    ref = _reference(st, Z, Mdir, 1.0, mt)
    assert bool(torch.isfinite(got).all())
    assert (got - ref).abs().max().item() <= 1e-10 * max(1.0, ref.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["xor_tanh_mlp", "resnet_bn_res_stride2", "stem_maxpool", "resnet_fast_kernels"])
def test_shared_direction_pairing_on_the_hip_engine(name):
    from lip_amd.engine import LinearizedNet
    if name == "resnet_fast_kernels":
        net, Z, mt = ResNet1M(4, input_shape=(8, 8, 3), widths=(16, 32), blocks_per_stage=1), \
            torch.rand(3, 8, 8, 3, dtype=F64, generator=torch.Generator().manual_seed(4)), "classifier"
    else:
        net, Z, mt = _cases()[name]
    st = create_state(net, 5, dtype=F64, logvar=-0.4)
    n = Z.shape[0]
    eng = LinearizedNet(st.to(device="cuda", dtype=torch.float32), Z.cuda().float(), mt, workspace_bytes=1 << 28, max_chunk=16)
    T = 7
    g = torch.Generator().manual_seed(9)
    Udir = torch.randn(T, eng.D, dtype=F64, generator=g)
    Xw = torch.randn(n, T, eng.K, dtype=F64, generator=g)
    got = so.input_grad_of_rank_one_terms(so.EngineExecutor(eng), [(Udir.cuda().float(), Xw.permute(1, 0, 2).reshape(T, -1).cuda())],
                                          1.0, mt, max_directions=4).double().cpu()
    ref = _reference_shared(st, Z, Udir, Xw, 1.0, mt)
    assert (got - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item()), (got - ref).abs().max().item()
