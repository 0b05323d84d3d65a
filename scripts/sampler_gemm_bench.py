"""The GEMM passes of a block of 256 posterior draws at the bench's CIFAR shapes (D = 1 084 586, r = 450 rows of the
orthonormalised factor): pass 1 lip_gemm_nt, the float64 stiff pass lip_dot_nt_f64, pass 2 lip_gemm_nn_axpy against
torch.addmm (hipBLASLt) on the same operands."""
import sys, time, torch
sys.path.insert(0, '.')
import lip_amd
from lip_amd import krylov
D, r, S = 1084586, 450, 256
g = torch.Generator().manual_seed(0)
Qm = torch.randn(r, D, generator=g).cuda() / 1000
V = krylov.fill_normal(S, D, 3)
T = torch.randn(S, r, generator=g).cuda()

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

fl = 2.0 * S * r * D
t1 = timed(lambda: krylov.gemm_nt(V, Qm))
t2 = timed(lambda: krylov.dot_nt(V, Qm[:96]))
out = torch.empty_like(V)
t3 = timed(lambda: krylov.gemm_nn_axpy(T, Qm, V, 14.0, out=out))
t4 = timed(lambda: torch.addmm(V, T, Qm, beta=14.0, out=out))
ref = torch.addmm(V, T, Qm, beta=14.0)
mine = krylov.gemm_nn_axpy(T, Qm, V, 14.0)
print(f"pass 1 lip_gemm_nt        {t1:7.3f} ms {fl / t1 / 1e9:6.1f} TF")
print(f"stiff  lip_dot_nt_f64 (96) {t2:7.3f} ms {2.0 * S * 96 * D / t2 / 1e9:6.1f} TF (f64)")
print(f"pass 2 lip_gemm_nn_axpy   {t3:7.3f} ms {fl / t3 / 1e9:6.1f} TF")
print(f"pass 2 torch.addmm        {t4:7.3f} ms {fl / t4 / 1e9:6.1f} TF   max diff {((mine - ref).abs().max() / ref.abs().max()).item():.2e}")
Vc = V.clone()
t5 = timed(lambda: krylov.gemm_nn_axpy(T, Qm, Vc, 1.0, out=Vc))
t6 = timed(lambda: torch.addmm(Vc, T, Qm, beta=1.0, out=Vc))
print(f"pass 2 in place: lip_gemm_nn_axpy {t5:7.3f} ms, torch.addmm {t6:7.3f} ms")
