"""GGN-vector products at P = 1, 2, 4, 8 on the CIFAR binding (the reference's one-vector call shape): ms per call."""
import sys, time, torch
sys.path.insert(0, '.')
import lip_amd
from lip_amd import krylov
from lip_amd.engine import LinearizedNet
from lip_amd.scalemodels import ResNet1M
from lip_amd.toymodels import create_state

net = ResNet1M(10); st = create_state(net, seed=1, dtype=torch.float32)
eng = LinearizedNet(st, torch.rand(50, 32, 32, 3).cuda(), "classifier", workspace_bytes=4 << 30, max_chunk=16)
for P in (1, 2, 4, 8, 16, 32):
    V = krylov.fill_rademacher(P, eng.D, 1, "cuda")
    out = torch.empty_like(V)
    for _ in range(5):
        eng.ggn_vp(V, 1.0, 0.0, out=out)
    torch.cuda.synchronize()
    t = time.perf_counter()
    reps = 50
    for _ in range(reps):
        eng.ggn_vp(V, 1.0, 0.0, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    print(f"P = {P}: {dt * 1e3:.3f} ms per call, {P / dt:.0f} GGN-vp/s, checksum {out.double().abs().sum().item():.9e}")
