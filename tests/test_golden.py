"""Committed golden vectors (tests/golden/oracle_vectors.npz, made by tests/golden/make_golden.py from the
float64 oracle): the oracle must keep reproducing them bit-for-bit-ish (CPU), the HIP path to fp32 tolerance
(GPU).  Nothing here reads /root/reference."""
import os

import numpy as np
import pytest
import torch

from fixtures import (make_classification_2d_data, make_classifier_state, make_sine_batch, make_toyregressor_state)
from impl import cpu64, impl  # noqa: F401
from lip_amd.scalemodels import ResNet1M
from lip_amd.toymodels import create_state

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_vectors.npz"))
T = lambda k: torch.from_numpy(GOLD[k])


def _close(impl, a, ref):
    a = cpu64(a).reshape(ref.shape)
    tol = impl.tol(1e-11, 2e-4)
    assert torch.allclose(a, ref, rtol=tol, atol=tol * ref.abs().max().item()), (a - ref).abs().max()


def test_sine_regressor_vectors(impl):
    X, _ = make_sine_batch()
    st = impl.state(make_toyregressor_state())
    V = impl.tensor(T("sine_V"))
    vp = impl.ggn.compute_ggn_vp(st, impl.tensor(X), "regressor", full_set_size=270)
    _close(impl, impl.rows(vp, V), T("sine_ggn_vp"))
    _, WTf = impl.ggn.compute_W_vps(st, impl.tensor(X), "regressor")
    _close(impl, impl.rows(WTf, V), T("sine_WT"))


def test_xor_classifier_vectors(impl):
    Xc, _ = make_classification_2d_data()
    Xc = Xc[::10]
    st = impl.state(make_classifier_state())
    V = impl.tensor(T("xor_V"))
    vp = impl.ggn.compute_ggn_vp(st, impl.tensor(Xc), "classifier", full_set_size=200)
    _close(impl, impl.rows(vp, V), T("xor_ggn_vp"))
    Wf, WTf = impl.ggn.compute_W_vps(st, impl.tensor(Xc), "classifier")
    _close(impl, impl.rows(WTf, V), T("xor_WT"))
    _close(impl, impl.rows(Wf, impl.tensor(T("xor_U"))), T("xor_W"))


def test_resnet_vectors(impl):
    net = ResNet1M(10, input_shape=(8, 8, 3), widths=(8, 16, 32), blocks_per_stage=1)
    st = impl.state(create_state(net, 77, dtype=torch.float64))
    vp = impl.ggn.compute_ggn_vp(st, impl.tensor(T("resnet_Z")), "classifier", full_set_size=30)
    _close(impl, impl.rows(vp, impl.tensor(T("resnet_V"))), T("resnet_ggn_vp"))
