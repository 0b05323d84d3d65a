"""Toy workloads — the nets of ``src/toymodels.py`` as layer programs.

``SimpleRegressor(numh, numl)``: ``numl`` x [Dense(numh) -> GELU] -> Dense(1), plus a scalar
``logvar`` collection that is *not* part of theta (``src/toymodels.py:4-24``).
``SimpleClassifier(numh, numl, numc)``: ``numl`` x [Dense(numh) -> tanh] -> Dense(numc)
(``src/toymodels.py:27-37``).
"""
from __future__ import annotations

import torch

from .netspec import NetSpec
from .utils import TrainState


def SimpleRegressor(numh: int, numl: int, in_features: int = 1) -> NetSpec:
    net = NetSpec((in_features,))
    t = 0
    for j in range(numl):
        t = net.dense(t, f"Dense_{j}", numh, act="gelu")
    net.dense(t, f"Dense_{numl}", 1)
    net.model_type = "regressor"
    return net


def SimpleClassifier(numh: int, numl: int, numc: int, in_features: int = 2) -> NetSpec:
    net = NetSpec((in_features,))
    t = 0
    for j in range(numl):
        t = net.dense(t, f"Dense_{j}", numh, act="tanh")
    net.dense(t, f"Dense_{numl}", numc)
    net.model_type = "classifier"
    return net


def LinearRegressor1D() -> NetSpec:
    """The hand-written linear 'model' of the reference's ``small_model_state`` fixture
    (``tests/fixtures.py:29-70``): mu = W*x + b with scalar W, b; flat order (W, b)."""
    net = NetSpec((1,))
    net.dense(0, "lin", 1)
    u = net.units[-1]
    u.kernel, u.bias = ("params", "W"), ("params", "b")
    net.model_type = "regressor"
    return net


def create_state(net: NetSpec, seed: int, dtype=torch.float32, logvar: float = 0.0,
                 randomize_bn: bool = True) -> TrainState:
    """Seeded-random-init ``state`` (the reference's checkpoints are absent)."""
    params, stats = net.init_params(seed, dtype=dtype, randomize_bn=randomize_bn)
    model_type = getattr(net, "model_type", "classifier")
    if model_type == "regressor":
        params["logvar"] = {"logvar": torch.tensor(float(logvar), dtype=dtype)}
    return TrainState(params=params, apply_fn=net.make_apply_fn(model_type, stats),
                      batch_stats=stats, net=net)
