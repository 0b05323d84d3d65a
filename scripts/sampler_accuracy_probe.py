"""Where does the eigh sampler's whitening error at the CIFAR config come from?  U = V Wm^T three ways."""
import json, math, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lip_amd  # noqa
from lip_amd import krylov
from lip_amd.ggn import get_engine
from lip_amd.sample import inv_matsqrt_vp
from lip_amd.scalemodels import ResNet1M
from lip_amd.toymodels import create_state

ALPHA, FULL, n, S = float(os.environ.get("ALPHA", "0.005")), 49000, 50, 8
dev = torch.device("cuda")
st = create_state(ResNet1M(10), seed=1231231234, dtype=torch.float32).to(device=dev, dtype=torch.float32)
Z = torch.rand(n, 32, 32, 3, generator=torch.Generator().manual_seed(280300)).to(dev)
eng = get_engine(st, Z, "classifier")
A = lambda B: eng.ggn_vp(B.contiguous(), FULL / n, ALPHA)
V = krylov.fill_normal(S, eng.D, 4242, dev)
op = inv_matsqrt_vp(st, Z, eng.D, ALPHA, "classifier", full_set_size=FULL, method="eigh")
pt = op.parts
Wm, Mc, a = pt.Wm, pt.Mc64.float(), 1.0 / math.sqrt(ALPHA)
Vd = V.double()
want = Vd @ Vd.T

def whiten(X):
    g = X.double() @ A(X).double().T
    return ((g - want).abs().max() / want.diagonal().max()).item()

def U_f32():
    return V @ Wm.T
def U_chunk(slab):
    D = Wm.shape[1]; body = D // slab * slab
    Vb = V[:, :body].reshape(S, body // slab, slab).permute(1, 0, 2)
    Wb = Wm[:, :body].reshape(Wm.shape[0], body // slab, slab).permute(1, 0, 2)
    U = torch.bmm(Vb, Wb.transpose(1, 2)).double().sum(0)
    return U + V[:, body:].double() @ Wm[:, body:].double().T
def U_f64():
    U = torch.zeros(S, Wm.shape[0], device=dev, dtype=torch.float64)
    for c in range(0, Wm.shape[1], 1 << 16):
        U += V[:, c:c + (1 << 16)].double() @ Wm[:, c:c + (1 << 16)].double().T
    return U
res = {}
U64 = U_f64()
for name, U in (("f32_gemm", U_f32()), ("chunk8192", U_chunk(8192)), ("chunk1024", U_chunk(1024)), ("chunk128", U_chunk(128)), ("f64", U64)):
    relU = ((U.double() - U64).norm() / U64.norm()).item()
    for second in ("f32", "f64"):
        if second == "f32":
            X = torch.addmm(V, (U.float() @ Mc).contiguous(), Wm, beta=a, alpha=1.0)
        else:
            X = (a * Vd + (U.double() @ pt.Mc64) @ Wm.double()).float()
        res[f"{name}|second={second}"] = dict(rel_err_U=relU, whitening=whiten(X))
# per-direction: eigenvectors of the factor Gram -> q_k = Wm^T u_k / sqrt(lam_k)
ev, Ug = torch.linalg.eigh(pt.WTW.double())
top = torch.argsort(ev, descending=True)[:5]
print(json.dumps(res, indent=1))
print("top eig beta*lam:", (pt.beta * ev[top]).tolist())
# the product path (orthonormalised factor)
t0 = time.perf_counter(); X = op.rows(V); torch.cuda.synchronize()
print("product path whitening:", whiten(X))
