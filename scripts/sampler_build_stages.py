"""Stages of a posterior-sampler build from a fresh (state, Z) binding at the bench's CIFAR config (synchronised)."""
import sys, time, math, torch
sys.path.insert(0, '.')
import lip_amd
from lip_amd import krylov, ggn, sample as smod
from lip_amd.scalemodels import ResNet1M
from lip_amd.toymodels import create_state

net = ResNet1M(10); st = create_state(net, seed=1, dtype=torch.float32)
st = st.replace(params=torch.utils._pytree.tree_map(lambda t: t.cuda(), st.params)) if hasattr(st, "replace") else st
Zw = torch.rand(50, 32, 32, 3).cuda(); Z = torch.rand(50, 32, 32, 3).cuda()
D = 1084586
smod.sample(st, Zw, D, 0.005, 1, "classifier", num_samples=200, full_set_size=49000)      # warm-up on another binding
smod._PARTS_CACHE.clear(); ggn.clear_engine_cache()
import gc; gc.collect(); torch.cuda.synchronize()
marks = []
def mark(name):
    torch.cuda.synchronize(); marks.append((name, time.perf_counter()))
mark("start")
W, WT = ggn.compute_W_vps(st, Z, "classifier"); eng = W.engine; mark("engine build + primal pass (compute_W_vps)")
Wm = ggn.materialize_factor(eng, 1.0); mark("factor rows (materialize_factor)")
G = ggn.gram_from_factor(Wm); G = torch.triu(G) + torch.triu(G, 1).T; mark("float64 Gram")
Gp, Gi, evp, Ug = smod._psd_and_pinv(G, smod.GRAM_RTOL, return_eig=True); mark("eigh + PSD projection + pseudo-inverse")
keep = evp > 0
kept = torch.nonzero(keep).flatten()
Qm = smod.orthonormal_factor(Wm, evp, Ug, kept); mark("orthonormalised factor Qm (float64 slabs)")
V = krylov.fill_normal(200, D, 3); mark("eps fill (200 draws)")
C = krylov.gemm_nt(V, Qm); C2 = krylov.dot_nt(V, Qm[:96]); mark("pass 1 (gemm_nt + dot_nt_f64)")
out = torch.addmm(V, C, Qm, beta=2.0); mark("pass 2 (addmm)")
for (n0, t0), (n1, t1) in zip(marks, marks[1:]):
    print(f"{n1:52s} {1e3 * (t1 - t0):8.2f} ms")
print(f"{'total':52s} {1e3 * (marks[-1][1] - marks[0][1]):8.2f} ms")
t = time.perf_counter(); smod._PARTS_CACHE.clear(); ggn.clear_engine_cache(); gc.collect(); torch.cuda.synchronize()
t = time.perf_counter(); S = smod.sample(st, Z, D, 0.005, 7, "classifier", num_samples=200, full_set_size=49000); torch.cuda.synchronize()
print(f"{'sample() end to end, fresh binding':52s} {1e3 * (time.perf_counter() - t):8.2f} ms")
