"""Timing of the float64-accumulated tall-skinny products and of the sampler's three passes at the CIFAR size."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lip_amd  # noqa
from lip_amd import krylov

D = 1084586
def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n
out = {}
for m, n in ((20, 20), (36, 36), (256, 96), (256, 32), (64, 64), (8, 450)):
    A, B = krylov.fill_normal(m, D, 1), krylov.fill_normal(n, D, 2)
    t = timeit(lambda: krylov.dot_nt(A, B))
    out[f"dot_nt {m}x{n}"] = dict(ms=t * 1e3, tflops_f64=2.0 * m * n * D / t / 1e12, GBps=4.0 * D * (m + n) / t / 1e9)
V = krylov.fill_normal(256, D, 3); Q = krylov.fill_normal(450, D, 4)
t = timeit(lambda: V @ Q.T); out["f32 gemm (256 x D)(450 x D)^T"] = dict(ms=t * 1e3, tflops=2.0 * 256 * 450 * D / t / 1e12)
t = timeit(lambda: krylov.gemm_nt(V, Q)); out["lip_gemm_nt (256 x D)(450 x D)^T"] = dict(ms=t * 1e3, tflops=2.0 * 256 * 450 * D / t / 1e12)
V8 = V[:8].contiguous()
t = timeit(lambda: krylov.gemm_nt(V8, Q)); out["lip_gemm_nt (8 x D)(450 x D)^T"] = dict(ms=t * 1e3, GBps=4.0 * D * 458 / t / 1e9)
t = timeit(lambda: V8 @ Q.T); out["f32 gemm (8 x D)(450 x D)^T"] = dict(ms=t * 1e3, GBps=4.0 * D * 458 / t / 1e9)
T = torch.randn(256, 450, device="cuda")
t = timeit(lambda: torch.addmm(V, T, Q, beta=2.0, alpha=1.0, out=V)); out["f32 addmm (256 x 450)(450 x D)"] = dict(ms=t * 1e3, tflops=2.0 * 256 * 450 * D / t / 1e12)
print(json.dumps(out, indent=1))
