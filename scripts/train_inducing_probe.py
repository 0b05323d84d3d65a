"""One inducing-point gradient step at the CIFAR config (n = 50 inducing images, data batch of 256): stage timing."""
import sys, time, torch
sys.path.insert(0, '.')
import lip_amd
from lip_amd.scalemodels import ResNet1M
from lip_amd.toymodels import create_state
from lip_amd import train_inducing as TI

net = ResNet1M(10)
st = create_state(net, seed=1, dtype=torch.float32).to(device='cuda', dtype=torch.float32)
Z = torch.rand(50, 32, 32, 3).cuda(); X = torch.rand(int(sys.argv[1]) if len(sys.argv) > 1 else 256, 32, 32, 3).cuda()
for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    loss, g = TI.variational_grad_scalable(Z, X, st, 0.005, model_type="classifier", full_set_size=49000, x_chunk=128)
    torch.cuda.synchronize(); print(f"rep {rep}: {time.perf_counter() - t:.3f} s  loss {loss:.4e}  |g| {g.norm().item():.4e} finite {torch.isfinite(g).all().item()}")
    Z = Z - 1e-3 * g / g.abs().max()
