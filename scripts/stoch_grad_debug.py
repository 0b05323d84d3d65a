"""Finiteness of every piece of the stochastic inducing-point gradient at the CIFAR config (50 inducing images, data batch
of 256, 256 probes, 2 x 40-step SLQ)."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lip_amd
from lip_amd import krylov, stochastic_grad as SG
from lip_amd.ggn import build_WTW, compute_W_vps
from lip_amd.lla import compute_curvature_approx
from lip_amd.scalemodels import ResNet1M
from lip_amd.toymodels import create_state
dev = torch.device("cuda", 0)
st = create_state(ResNet1M(10), seed=1231231234, dtype=torch.float32).to(device=dev, dtype=torch.float32)
Z = torch.rand(50, 32, 32, 3, generator=torch.Generator().manual_seed(280300)).to(dev)
X = torch.rand(int(sys.argv[1]) if len(sys.argv) > 1 else 64, 32, 32, 3, generator=torch.Generator().manual_seed(77)).to(dev)
alpha, N = 0.005, 49000
S_rows = compute_curvature_approx(st, X, alpha=alpha, model_type="classifier", full_set_size=N).rows
Wz, WzT = compute_W_vps(st, Z, model_type="classifier", full_set_size=None)
eng = Wz.engine; D = eng.D; inner = WzT.out_shape; d = math.prod(inner)
WTW = build_WTW(Wz, WzT, inner, d, dtype=torch.float64, block=1)
print("WTW finite", bool(torch.isfinite(WTW).all()), "max", WTW.abs().max().item())
WT_rows = lambda V: WzT.rows(V.contiguous()).reshape(V.shape[0], d)
W_rows = lambda Xs: Wz.rows(Xs.to(torch.float32).reshape((Xs.shape[0],) + inner).contiguous())
stn = int(sys.argv[2]) if len(sys.argv) > 2 else 48
probes = krylov.fill_rademacher(stn, D, 5, dev)
vec = SG.HipVec()
I = torch.eye(d, dtype=torch.float64, device=dev)
Minv = torch.linalg.inv(I / (N / 50) + WTW / alpha); Minv = 0.5 * (Minv + Minv.T)
print("Minv finite", bool(torch.isfinite(Minv).all()), "max", Minv.abs().max().item())
val, terms, info = SG._hutchpp_value_and_terms(S_rows, WT_rows, W_rows, Minv, alpha, probes, stn - 16, 16, vec)
print("hutch value", val, info, [(bool(torch.isfinite(U).all()), bool(torch.isfinite(Xx).all()), U.abs().max().item(), Xx.abs().max().item()) for U, Xx in terms])
for k in (4, 16, 40):
    v2, t2 = SG._slq_value_and_terms(WT_rows, W_rows, D, d, alpha, math.sqrt(N / 50), probes[:2].contiguous(), k, vec)
    print("slq k", k, "value", v2, "terms finite", all(bool(torch.isfinite(U).all()) and bool(torch.isfinite(Xx).all()) for U, Xx in t2),
          "max |U|", max(U.abs().max().item() for U, _ in t2), "max |X|", max(Xx.abs().max().item() for _, Xx in t2))
