"""BASELINE configs[4] slice on one GPU: full-resolution ResNet-50 (25.6 M parameters, K = 1000), n images x P probes."""
import sys, time, torch
sys.path.insert(0, '.')
import lip_amd
from lip_amd import _native as nv, krylov
from lip_amd.engine import LinearizedNet
from lip_amd.scalemodels import ResNet50
from lip_amd.toymodels import create_state

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
P = int(sys.argv[2]) if len(sys.argv) > 2 else 64
net = ResNet50(1000)
t = time.perf_counter(); st = create_state(net, seed=1, dtype=torch.float32); print("init", time.perf_counter() - t)
Z = torch.rand(n, 224, 224, 3).cuda()
t = time.perf_counter()
eng = LinearizedNet(st, Z, "classifier", workspace_bytes=64 << 30, max_chunk=P)
torch.cuda.synchronize(); print("engine build + primal", time.perf_counter() - t, "chunk", eng.chunk)
V = krylov.fill_rademacher(P, eng.D, 1, "cuda")
for _ in range(2):
    Y = eng.ggn_vp(V, 1.0, 0.0)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(3):
    Y = eng.ggn_vp(V, 1.0, 0.0)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
fl = eng.flops_per_probe()
print(f"n={n} P={P}: {dt*1e3:.1f} ms/step, {P/dt:.1f} GGN-vp/s (x{n} images), {sum(fl.values())*P/dt/1e12:.1f} TFLOP/s, finite {torch.isfinite(Y).all().item()}")
eng.profile(True); eng.ggn_vp(V, 1.0, 0.0); pr = eng.profile_read(); eng.profile(False)
for k, (ms, c) in sorted(pr.items(), key=lambda kv: -kv[1][0]):
    print(f"  kind {k}: {ms:.1f} ms in {c} launches" + (f"  {fl[k]*P/ms/1e9:.1f} TF" if k in fl else ""))
