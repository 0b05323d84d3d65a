"""Stability of the SLQ adjoint in float32: reverse mode through the re-orthogonalised Golub-Kahan recurrence (torch.autograd\nthrough oracle/matfree.py) against the adjoint of stochastic_grad.py, on a toy factor whose spectrum spans seven decades.\nUsage: python scripts/slq_adjoint_stability.py [k]"""
import sys, math, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import lip_amd
from lip_amd import stochastic_grad as SG
from oracle.matfree import bidiag, integrand_funm_product_logdet
torch.manual_seed(0)
D, d, k = 400, 40, int(sys.argv[1]) if len(sys.argv)>1 else 20
alpha = 0.005
U0,_ = torch.linalg.qr(torch.randn(D,d,dtype=torch.float64))
sv = torch.logspace(0, 3.5, d, dtype=torch.float64)        # singular values 1..3e3 -> eigenvalues up to 1e7
V0,_ = torch.linalg.qr(torch.randn(d,d,dtype=torch.float64))
W64 = (U0*sv)@V0.T                                           # D x d
p64 = torch.sign(torch.randn(2, D, dtype=torch.float64))
def autograd_grad(dt):
    W = W64.to(dt).clone().requires_grad_(True); sa = math.sqrt(alpha)
    A = lambda v: torch.cat([sa*v, W.T@v]); AT = lambda u: sa*u[:D] + W@u[D:]
    quad = integrand_funm_product_logdet(bidiag(k))
    val = torch.stack([quad(A, AT, p.to(dt)) for p in p64]).mean()
    g, = torch.autograd.grad(val, W)
    return float(val), g.double()
def plain_adjoint(dt):
    W = W64.to(dt)
    WT_rows = lambda V: V@W; W_rows = lambda X: X.to(dt)@W.T
    val, terms = SG._slq_value_and_terms(WT_rows, W_rows, D, d, alpha, 1.0, p64.to(dt), k, SG.TorchVec())
    G = sum(U.double().T @ X.double() for U,X in terms)
    return val, G
v64, g64 = autograd_grad(torch.float64)
for name, fn in (("autograd(reorth) f32", lambda: autograd_grad(torch.float32)), ("product adjoint f64", lambda: plain_adjoint(torch.float64)), ("product adjoint f32", lambda: plain_adjoint(torch.float32))):
    v, g = fn()
    print(f"{name}: value {v:.6e} (f64 {v64:.6e}) grad rel err {((g-g64).norm()/g64.norm()).item():.3e}  |g| {g.norm().item():.3e}")
