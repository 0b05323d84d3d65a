"""gradient of the inducing-point objectives w.r.t. Z (reference src/train_inducing.py:195-232) —
product (HIP factors + Gram algebra + torch.func input derivative) against reverse mode through the oracle's dense
float64 construction, and a few optimiser steps that must decrease the exact objective."""
import math

import pytest
import torch

from lip_amd.toymodels import SimpleClassifier, SimpleRegressor, create_state
from lip_amd.scalemodels import ResNet1M
from oracle import train_inducing as OT

F64 = torch.float64


def _case(name):
    g = torch.Generator().manual_seed(3)
    if name == "xor":
        net, Z, X, mt = SimpleClassifier(8, 2, 2), torch.randn(5, 2, dtype=F64, generator=g), torch.randn(12, 2, dtype=F64, generator=g), "classifier"
    elif name == "sine":
        net, Z, X, mt = SimpleRegressor(6, 2), torch.randn(4, 1, dtype=F64, generator=g), torch.randn(9, 1, dtype=F64, generator=g), "regressor"
    else:
        net = ResNet1M(3, input_shape=(6, 6, 3), widths=(4, 8), blocks_per_stage=1)
        Z, X, mt = torch.rand(3, 6, 6, 3, dtype=F64, generator=g), torch.rand(5, 6, 6, 3, dtype=F64, generator=g), "classifier"
    return create_state(net, 5, dtype=F64, logvar=-0.4), Z, X, mt


def test_oracle_gradient_matches_finite_differences():
    st, Z, X, mt = _case("xor")
    for obj in (OT.objective_dense_t, OT.objective_scalable_t):
        v, g = OT.variational_grad(obj, Z, X, st, 0.7, mt, 40)
        E = torch.randn(Z.shape, dtype=F64, generator=torch.Generator().manual_seed(1))
        h = 1e-5
        fd = (float(obj(Z + h * E, X, st, 0.7, mt, 40)) - float(obj(Z - h * E, X, st, 0.7, mt, 40))) / (2 * h)
        assert abs(fd - float((g * E).sum())) <= 1e-6 * max(1.0, abs(fd))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["xor", "sine", "resnet"])
def test_variational_grads_match_oracle(name):
    from lip_amd.train_inducing import variational_grad_dense, variational_grad_scalable
    st, Z, X, mt = _case(name)
    alpha, N = 0.7, 40
    for fn, obj in ((variational_grad_dense, OT.objective_dense_t), (variational_grad_scalable, OT.objective_scalable_t)):
        v_o, g_o = OT.variational_grad(obj, Z, X, st, alpha, mt, N)
        v, g = fn(Z.cuda(), X.cuda(), st, alpha, key=0, model_type=mt, full_set_size=N)
        g = g.double().cpu()
        assert g.shape == Z.shape
        # float32 factors / float64 small algebra; tolerance 2e-3 of the gradient's scale
        assert (g - g_o).abs().max().item() <= 2e-3 * g_o.abs().max().item(), (name, fn.__name__, (g - g_o).abs().max().item(), g_o.abs().max().item())
        assert abs(v - v_o) <= 2e-3 * max(1.0, abs(v_o)), (name, fn.__name__, v, v_o)
    # chunked data batch gives the same gradient
    v2, g2 = variational_grad_scalable(Z.cuda(), X.cuda(), st, alpha, key=0, model_type=mt, full_set_size=N, x_chunk=2)
    assert (g2.double().cpu() - g_o).abs().max().item() <= 2e-3 * g_o.abs().max().item()
    assert abs(v2 - v) <= 1e-4 * max(1.0, abs(v))


@pytest.mark.gpu
def test_value_differences_match_the_oracle_objective():
    """The product drops Z-independent constants (reference :68,82): compare differences between two Z."""
    from lip_amd.train_inducing import variational_grad_scalable
    st, Z, X, mt = _case("xor")
    Z2 = Z + 0.3
    a = variational_grad_scalable(Z.cuda(), X.cuda(), st, 0.7, model_type=mt, full_set_size=40)[0]
    b = variational_grad_scalable(Z2.cuda(), X.cuda(), st, 0.7, model_type=mt, full_set_size=40)[0]
    ao = float(OT.objective_scalable_t(Z, X, st, 0.7, mt, 40)); bo = float(OT.objective_scalable_t(Z2, X, st, 0.7, mt, 40))
    assert abs((a - b) - (ao - bo)) <= 1e-3 * max(1.0, abs(ao - bo))


@pytest.mark.gpu
def test_inducing_point_training_decreases_the_objective():
    from lip_amd.train_inducing import AdamW, train_inducing_points, variational_grad_scalable
    st, Z, X, mt = _case("xor")
    Z0 = Z.float().cuda()
    l0 = variational_grad_scalable(Z0, X.cuda(), st, 0.7, model_type=mt, full_set_size=40)[0]
    Zt, losses = train_inducing_points(st, Z0, AdamW(lr=5e-2, weight_decay=0.0), [X.float().cuda()], mt, alpha=0.7,
                                       num_steps=15, full_set_size=40, scalable=True)
    l1 = variational_grad_scalable(Zt, X.cuda(), st, 0.7, model_type=mt, full_set_size=40)[0]
    assert l1 < l0 and math.isfinite(l1)
