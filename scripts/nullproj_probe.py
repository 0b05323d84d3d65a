"""test_nullproj's literal CG route under the current build: residual in the GGN kernel relative to the scale, CG
iterations, and the same for slightly perturbed right-hand sides (is the outcome a property of the build or chaos of a
float32 CG on a cond-1e17 Gram?)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lip_amd  # noqa
from lip_amd import krylov
import src.ggn as hg
from fixtures import make_sine_batch, make_toyregressor_state
X, y = make_sine_batch()
st = make_toyregressor_state().to(device="cuda", dtype=torch.float32)
Xd = X.float().cuda()
D = 241
Wfun, WTfun = hg.compute_W_vps(st, Xd, "regressor")
comp = lambda U: WTfun.rows(Wfun.rows(U))
for seed, eps in ((41234, 0.0), (41234, 1e-6), (41234, 1e-5), (1, 0.0), (2, 0.0), (3, 0.0)):
    v = (torch.randn(D, dtype=torch.float64, generator=torch.Generator().manual_seed(seed)) * 10).float().cuda()
    if eps:
        v = v * (1 + eps * torch.randn(D, generator=torch.Generator().manual_seed(7)).cuda())
    x, info = krylov.cg(comp, WTfun(v)[None].contiguous())
    full = v - Wfun(x[0])
    resid = Wfun(WTfun(full)).double().abs().max().item()
    scale = Wfun(WTfun(v)).double().abs().max().item()
    print(f"seed {seed} eps {eps:g}: resid/scale {resid / scale:.3e}  iterations {info['iterations']}  |r| {info['residual_norm'].max().item():.3e}")
print("maxiter sweep (seeds 41234, 1, 2, 3, 4, 5):")
for mi in (16, 24, 32, 48, 80, 160):
    out = []
    for seed in (41234, 1, 2, 3, 4, 5):
        v = (torch.randn(D, dtype=torch.float64, generator=torch.Generator().manual_seed(seed)) * 10).float().cuda()
        x, info = krylov.cg(comp, WTfun(v)[None].contiguous(), maxiter=mi)
        full = v - Wfun(x[0])
        out.append(Wfun(WTfun(full)).double().abs().max().item() / Wfun(WTfun(v)).double().abs().max().item())
    print(f"seed maxiter {mi}: " + " ".join(f"{o:.2e}" for o in out))
