"""Generates tests/golden/oracle_vectors.npz from the float64 CPU oracle (the reference cannot run here:
jax / flax / matfree are absent, SURVEY G1 — so these are the oracle's own outputs, pinned by the
reference's RNG-free known answers in tests/test_ggn.py etc.).  Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import lip_amd  # noqa: E402,F401
from fixtures import (make_classification_2d_data, make_classifier_state, make_sine_batch,  # noqa: E402
                      make_toyregressor_state)
from lip_amd.scalemodels import ResNet1M  # noqa: E402
from lip_amd.toymodels import create_state  # noqa: E402
import oracle.ggn as og  # noqa: E402

F64 = torch.float64
out = {}
g = torch.Generator().manual_seed(2024)

X, _ = make_sine_batch()
st = make_toyregressor_state()
V = torch.randn(3, 241, dtype=F64, generator=g)
vp = og.compute_ggn_vp(st, X, "regressor", full_set_size=270)
Wf, WTf = og.compute_W_vps(st, X, "regressor")
out["sine_V"] = V.numpy()
out["sine_ggn_vp"] = torch.stack([vp(v) for v in V]).numpy()
out["sine_WT"] = torch.stack([WTf(v) for v in V]).numpy()

Xc, _ = make_classification_2d_data()
Xc = Xc[::10]
stc = make_classifier_state()
Vc = torch.randn(3, 354, dtype=F64, generator=g)
vpc = og.compute_ggn_vp(stc, Xc, "classifier", full_set_size=200)
Wc, WTc = og.compute_W_vps(stc, Xc, "classifier")
Uc = torch.randn(2, 20, 2, dtype=F64, generator=g)
out["xor_V"] = Vc.numpy()
out["xor_ggn_vp"] = torch.stack([vpc(v) for v in Vc]).numpy()
out["xor_WT"] = torch.stack([WTc(v) for v in Vc]).numpy()
out["xor_U"] = Uc.numpy()
out["xor_W"] = torch.stack([Wc(u) for u in Uc]).numpy()

net = ResNet1M(10, input_shape=(8, 8, 3), widths=(8, 16, 32), blocks_per_stage=1)
str_ = create_state(net, 77, dtype=F64)
Zr = torch.rand(3, 8, 8, 3, dtype=F64, generator=g)
D = sum(p.numel() for p in _ for _ in []) if False else None
from lip_amd.utils import flatten_nn_params  # noqa: E402
Dr = flatten_nn_params(str_.params)[0].numel()
Vr = torch.randn(2, Dr, dtype=F64, generator=g)
vpr = og.compute_ggn_vp(str_, Zr, "classifier", full_set_size=30)
out["resnet_Z"] = Zr.numpy()
out["resnet_V"] = Vr.numpy()
out["resnet_ggn_vp"] = torch.stack([vpr(v) for v in Vr]).numpy()

np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_vectors.npz"), **out)
print({k: v.shape for k, v in out.items()})
