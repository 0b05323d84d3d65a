"""Stage timing of the posterior sampler at the CIFAR config (n = 50, K = 10, D = 1.08 M)."""
import sys, time, math, torch
sys.path.insert(0, '.')
import lip_amd
from lip_amd.scalemodels import ResNet1M
from lip_amd.toymodels import create_state
from lip_amd import sample as S, krylov
from lip_amd.ggn import compute_W_vps, materialize_factor, gram_from_factor

def T(name, f):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize()
    print(f"{name:28s} {1e3 * (time.perf_counter() - t):8.1f} ms"); return r

net = ResNet1M(10)
st = create_state(net, seed=1231231234, dtype=torch.float32).to(device='cuda', dtype=torch.float32)
Z = torch.rand(50, 32, 32, 3, generator=torch.Generator().manual_seed(280300)).cuda()
for rep in range(2):
    print("--- rep", rep)
    S._PARTS_CACHE.clear()
    from lip_amd import ggn as G
    if rep == 0:
        Wf, WTf = T("engine build + primal", lambda: compute_W_vps(st, Z, "classifier"))
    eng = Wf.engine
    Wm = T("materialize_factor", lambda: materialize_factor(eng, 1.0))
    G64 = T("gram f64", lambda: gram_from_factor(Wm))
    Gp = T("psd + pinv (eigh f64)", lambda: S._psd_and_pinv(G64))
    A64 = 0.005 * torch.eye(500, device='cuda', dtype=torch.float64) + 980.0 * Gp[0]
    fA = T("f(A) (eigh f64)", lambda: krylov.dense_funm_sym_eigh(lambda x: 1 / torch.sqrt(x), None, floor=0.005)(A64))
    Eps = T("fill_normal 200", lambda: krylov.fill_normal(200, eng.D, 5, 'cuda'))
    U = T("Eps @ Wm^T", lambda: Eps @ Wm.T)
    X = T("small algebra", lambda: (U @ fA.float()) @ Gp[1].float() - U @ Gp[1].float())
    O = T("X @ Wm", lambda: X @ Wm)
    T("sample() S=200 end-to-end", lambda: S.sample(st, Z, eng.D, 0.005, 7 + rep, "classifier", num_samples=200, full_set_size=49000))
    T("sample() S=200 cached", lambda: S.sample(st, Z, eng.D, 0.005, 9 + rep, "classifier", num_samples=200, full_set_size=49000))
