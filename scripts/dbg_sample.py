import sys, torch, math
sys.path.insert(0, '.')
import lip_amd
from lip_amd.scalemodels import ResNet1M
from lip_amd.toymodels import create_state
from lip_amd import sample as S, krylov
net = ResNet1M(10)
st = create_state(net, seed=1231231234, dtype=torch.float32).to(device='cuda', dtype=torch.float32)
Z = torch.rand(50, 32, 32, 3, generator=torch.Generator().manual_seed(280300)).cuda()
parts = S._SamplerParts(st, Z, 1084586, 0.005, "classifier", 49000, None, "lanczos")
print("WTW finite", torch.isfinite(parts.WTW).all().item())
ev = torch.linalg.eigvalsh(parts.WTW.double()); print('WTW eig min/max', ev.min().item(), ev.max().item(), 'n>1e-6max', (ev > 1e-6*ev.max()).sum().item())
print('Gpinv finite', torch.isfinite(parts.G_pinv).all().item(), parts.G_pinv.abs().max().item())
Eps = krylov.fill_normal(4, parts.eng.D, 5, 'cuda')
U = parts.WTfun.rows(Eps).reshape(4, parts.d)
print('U finite', torch.isfinite(U).all().item(), U.abs().max().item())
print('depth', parts.depth, 'beta', parts.beta)
Q, diag, off = krylov.lanczos_tridiag(lambda X: X @ parts.A_d, U.contiguous(), parts.depth)
print('Q finite', torch.isfinite(Q).all().item(), 'diag finite', torch.isfinite(diag).all().item(), 'off finite', torch.isfinite(off).all().item())
print('diag', diag[0,:6], diag[0,-4:]); print('off', off[0,:6], off[0,-4:])
fU = parts.f_small(U); print('fU finite', torch.isfinite(fU).all().item())
parts2 = S._SamplerParts(st, Z, 1084586, 0.005, "classifier", 49000, None, "eigh")
fU2 = parts2.f_small(U); print('eigh fU finite', torch.isfinite(fU2).all().item(), 'rel diff', ((fU-fU2).norm()/fU2.norm()).item())
out = parts2.apply(Eps); print('out finite', torch.isfinite(out).all().item(), out.std().item(), 1/math.sqrt(0.005))
