"""PyTorch restatement of the reference's ``tests/fixtures.py`` (same geometry and constants;
the RNG streams are torch's, and MAP weights are seeded random inits because the reference's
``checkpoint/`` directory is git-ignored and absent — SURVEY G3)."""
import os

import numpy as np
import pytest
import torch

from lip_amd.toymodels import (LinearRegressor1D, SimpleClassifier, SimpleRegressor, create_state)
from lip_amd.utils import TrainState

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
F64 = torch.float64


def _randn(seed, shape, dtype=F64):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed), dtype=dtype)


def make_regression_1d_data(dtype=F64):
    """reference tests/fixtures.py:18-26"""
    X = torch.tensor([[-1.0], [0.0], [1.1], [3.5]], dtype=dtype)
    y = 2.0 * X + 0.1 * _randn(42, X.shape, dtype)
    return X, y


def make_small_model_state(dtype=F64) -> TrainState:
    """reference tests/fixtures.py:29-70 — mu = W x + b, scalar W, b; flat order (W, b)."""
    net = LinearRegressor1D()
    g = torch.Generator().manual_seed(0)
    W = 0.1 * torch.randn((), generator=g, dtype=F64)
    b = 0.1 * torch.randn((), generator=g, dtype=F64)
    logvar = 0.1 * torch.rand((), generator=g, dtype=F64)
    params = {"params": {"W": W.reshape(1, 1).to(dtype), "b": b.reshape(1).to(dtype)},
              "logvar": {"logvar": logvar.to(dtype)}}
    return TrainState(params=params, apply_fn=net.make_apply_fn("regressor"), net=net)


def make_toyregressor_state(dtype=F64) -> TrainState:
    """reference tests/fixtures.py:73-97 — SimpleRegressor(8, 4) (config/toy/toyregressor_sine.yml)."""
    return create_state(SimpleRegressor(8, 4), seed=1234, dtype=dtype, logvar=-1.0)


def make_sine_batch(dtype=F64):
    """reference tests/fixtures.py:100-119: 90/10 split, batch 16, no shuffle, drop_last ->
    the first test batch is rows 270:286 of data/sine.npz."""
    d = np.load(os.path.join(GOLDEN, "sine.npz"))
    x, y = torch.from_numpy(d["x"]).to(dtype), torch.from_numpy(d["y"]).to(dtype)
    return x[270:286], y[270:286]


def make_classification_2d_data(dtype=F64):
    """reference tests/fixtures.py:122-147: two blobs of 100 points at (-1,0), (1,0), sigma 0.5."""
    c0 = _randn(1, (100, 2)) * 0.5 + torch.tensor([-1.0, 0.0], dtype=F64)
    c1 = _randn(2, (100, 2)) * 0.5 + torch.tensor([1.0, 0.0], dtype=F64)
    X = torch.cat([c0, c1]).to(dtype)
    y = torch.cat([torch.zeros(100), torch.ones(100)]).to(torch.int64)
    return X, y


def make_classifier_state(dtype=F64) -> TrainState:
    """reference tests/fixtures.py:150-190 — SimpleClassifier(16, 2, 2) (config/toy/toyclassifier_xor.yml)."""
    return create_state(SimpleClassifier(16, 2, 2), seed=12345, dtype=dtype)


def make_matrix_test_suite(dtype=F64, n3=3000):
    """reference tests/fixtures.py:193-213"""
    M1 = torch.diag(torch.tensor([1., 2., 3.], dtype=dtype))
    M2 = torch.tensor([[1., 4, 50], [-30, 4., 16], [12, 6, 5.]], dtype=dtype)
    M2 = M2 @ M2.T
    G = _randn(45895, (n3, n3)).to(dtype)
    return M1, M2, G @ G.T


@pytest.fixture
def regression_1d_data():
    return make_regression_1d_data()


@pytest.fixture
def small_model_state():
    return make_small_model_state()


@pytest.fixture
def toyregressor_state():
    return make_toyregressor_state()


@pytest.fixture
def sine_data():
    return make_sine_batch()


@pytest.fixture
def classification_2d_data():
    return make_classification_2d_data()


@pytest.fixture
def classifier_state():
    return make_classifier_state()


@pytest.fixture(scope="module")
def matrix_test_suite():
    return make_matrix_test_suite()
