"""Deflated float32 CG on (GGN + alpha I) x = b at the CIFAR config (alpha = 0.005) with the stopping tolerance at the noise
floor of the deflated product (1e-3) and at three times it, repeated: iteration counts and forward errors against the closed
form.  Argument `private` gives the engine its own workspace instead of the shared pool (same picture)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lip_amd
from lip_amd import krylov
from lip_amd.ggn import get_engine, clear_engine_cache
from lip_amd.engine import LinearizedNet
from lip_amd.sample import range_deflation
from lip_amd.scalemodels import ResNet1M
from lip_amd.toymodels import create_state
ALPHA, FULL, N_IMG, S = 0.005, 49000, 50, 8
dev = torch.device("cuda")
st = create_state(ResNet1M(10), seed=1231231234, dtype=torch.float32).to(device=dev, dtype=torch.float32)
Z = torch.rand(N_IMG, 32, 32, 3, generator=torch.Generator().manual_seed(280300)).to(dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "shared"
eng = get_engine(st, Z, "classifier", workspace_bytes=(8 << 30) if mode == "private" else None)
print("mode", mode, "chunk", eng.chunk)
scale = FULL / N_IMG
A = lambda B: eng.ggn_vp(B.contiguous(), scale, ALPHA)
V = krylov.fill_normal(S, eng.D, 4242, dev)
B = V[:8].contiguous()
defl = range_deflation(st, Z, eng.D, ALPHA, "classifier", FULL)
Xref = defl.closed_form(B, lambda lam: 1.0 / lam, ALPHA)
for stall in (3, 3, 2, 1):
    for tol in (1e-3, 3e-3):
        X, info = krylov.cg_deflated(A, B, defl, tol=tol, maxiter=50, stall=stall)
        err = ((X - Xref).norm(dim=1) / Xref.norm(dim=1))
        print(f"stall {stall} tol {tol}: iterations {info['iterations']} err max {err.max().item():.3e} all {[round(float(e), 4) for e in err]}")
