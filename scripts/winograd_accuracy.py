"""Accuracy of the Winograd route against the direct route and the float64 oracle at the bench geometry (ResNet1M,
n = 50, seeds of tests/test_surface_extra.py), and the eigenpair residual of tests/test_sampler_fullsize.py on both."""
import math, sys, torch
sys.path.insert(0, '.')
import lip_amd
from lip_amd import _native as nv, krylov
from lip_amd.scalemodels import ResNet1M
from lip_amd.toymodels import create_state
import oracle.ggn as og
import src.ggn as hg

lib = nv.load()
net = ResNet1M(10)
st64 = create_state(net, 1231231234, dtype=torch.float64)
Z = torch.rand(50, 32, 32, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(280300))
vp = hg.compute_ggn_vp(st64.to(device="cuda", dtype=torch.float32), Z.cuda().float(), "classifier", full_set_size=49000)
V = krylov.fill_rademacher(8, vp.engine.D, 3, "cuda")
ref = og.compute_ggn_vp_batched(st64, Z, "classifier", full_set_size=49000)(V[0].double().cpu())
out = {}
for mode in (0, 1):
    lib.lip_set_winograd(mode)
    out[mode] = vp(V).double().cpu()
    e = ((out[mode][0] - ref).abs().max() / ref.abs().max()).item()
    e2 = ((out[mode][0] - ref).norm() / ref.norm()).item()
    print(f"winograd={mode}: vs float64 oracle max-rel {e:.3e}  l2-rel {e2:.3e}")
d = ((out[1] - out[0]).abs().max() / out[0].abs().max()).item()
print(f"winograd vs direct: max-rel {d:.3e}  l2-rel {((out[1]-out[0]).norm()/out[0].norm()).item():.3e}")
# J v alone (tangent tape) and J^T u alone (backward tape)
eng = vp.engine
for mode in (0, 1):
    lib.lip_set_winograd(mode)
    out[("j", mode)] = eng.jvp(V).double().cpu()
    U = torch.randn(8, eng.n, eng.K, generator=torch.Generator().manual_seed(1)).cuda()
    out[("t", mode)] = eng.vjp(U).double().cpu()
for k in ("j", "t"):
    a, b = out[(k, 0)], out[(k, 1)]
    print(f"{'J v' if k == 'j' else 'J^T u'}: winograd vs direct max-rel {((a-b).abs().max()/a.abs().max()).item():.3e}  l2-rel {((a-b).norm()/a.norm()).item():.3e}")
