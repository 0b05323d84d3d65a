"""Timing of the second-order pass alone at the CIFAR config (50 inducing images, K = 10 directions each)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lip_amd  # noqa
from lip_amd.scalemodels import ResNet1M
from lip_amd.toymodels import create_state
from lip_amd import train_inducing as TI

net = ResNet1M(10)
st = create_state(net, seed=1, dtype=torch.float32).to(device='cuda', dtype=torch.float32)
Z = torch.rand(50, 32, 32, 3).cuda()
from lip_amd.ggn import get_engine
eng = get_engine(st, Z, "classifier")
M = torch.randn(500, eng.D, device="cuda")
for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    g = TI._input_grad_of_pairing(st, Z, M, "classifier")
    torch.cuda.synchronize(); print(f"second-order pass rep {rep}: {time.perf_counter() - t:.3f} s  finite {torch.isfinite(g).all().item()}")
