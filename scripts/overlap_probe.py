"""Does f32-MFMA work overlap with HBM streaming on MI355X?  Runs the MFMA-only ablation of the implicit-GEMM
kernels (LIP_ABLATE=6 must be set) alone, a large device copy alone, and both concurrently on two streams."""
import sys, time, torch
sys.path.insert(0, '.')
import lip_amd
from lip_amd import krylov
from lip_amd.engine import LinearizedNet
from lip_amd.scalemodels import ResNet1M
from lip_amd.toymodels import create_state
net = ResNet1M(10)
st = create_state(net, seed=1, dtype=torch.float32)
Z = torch.rand(50, 32, 32, 3).cuda()
eng = LinearizedNet(st, Z, "classifier", workspace_bytes=24 << 30, max_chunk=256)
V = krylov.fill_rademacher(256, eng.D, 1, "cuda")
Y = torch.empty_like(V)
a = torch.empty(1 << 29, device="cuda")   # 2 GiB
b = torch.empty_like(a)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run_mfma(n):
    with torch.cuda.stream(s1):
        for _ in range(n): eng.ggn_vp(V, 1.0, 0.0, out=Y)
def run_copy(n):
    with torch.cuda.stream(s2):
        for _ in range(n): b.copy_(a)
def timed(f):
    torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); return time.perf_counter() - t
run_mfma(1); run_copy(2)
t_m = timed(lambda: run_mfma(3))
ncopy = 120
t_c = timed(lambda: run_copy(ncopy))
t_b = timed(lambda: (run_mfma(3), run_copy(ncopy)))
print(f"mfma alone {t_m*1e3:.1f} ms | copy alone {t_c*1e3:.1f} ms ({ncopy*2*a.numel()*4/t_c/1e12:.2f} TB/s) | both {t_b*1e3:.1f} ms  (max {max(t_m,t_c)*1e3:.1f}, sum {(t_m+t_c)*1e3:.1f})")
