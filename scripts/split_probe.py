"""Accuracy of the split-precision (bf16x3) MFMA path vs the float64 oracle and vs the f32 path."""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import lip_amd
from lip_amd import _native as nv, krylov
from lip_amd.engine import LinearizedNet
from lip_amd.scalemodels import ResNet1M
from lip_amd.toymodels import create_state
import oracle.ggn as og
lib = nv.load()
net = ResNet1M(10, input_shape=(16, 16, 3), widths=(32, 64, 128), blocks_per_stage=1)
st64 = create_state(net, 3, dtype=torch.float64)
Z = torch.rand(6, 16, 16, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(0))
eng = LinearizedNet(st64.to(device="cuda", dtype=torch.float32), Z.cuda().float(), "classifier")
V = krylov.fill_rademacher(4, eng.D, 1, "cuda")
ref_vp = og.compute_ggn_vp_batched(st64, Z, "classifier", full_set_size=60)
ref = torch.stack([ref_vp(v) for v in V.double().cpu()])
for mode in (0, 1):
    nv.check(lib.lip_set_precision(mode))
    Y = eng.ggn_vp(V, 10.0, 0.0).double().cpu()
    print("precision", mode, "rel max err", ((Y - ref).abs().max() / ref.abs().max()).item(), "rel l2", ((Y - ref).norm() / ref.norm()).item())
