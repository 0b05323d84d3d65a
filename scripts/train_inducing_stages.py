"""Stage timing of one inducing-point gradient step at the CIFAR config (wraps the module's helpers with timers)."""
import os, sys, time, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lip_amd  # noqa
from lip_amd.scalemodels import ResNet1M
from lip_amd.toymodels import create_state
from lip_amd import train_inducing as TI, ggn as G

acc = collections.defaultdict(float)
def timed(mod, name):
    f = getattr(mod, name)
    def w(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); acc[name] += time.perf_counter() - t
        return r
    setattr(mod, name, w)
for m, nm in ((TI, "_factor_of"), (TI, "_gram64"), (TI, "_input_grad_of_pairing"), (G, "gram_from_factor"), (G, "materialize_factor"), (G, "get_engine")):
    timed(m, nm)
net = ResNet1M(10)
st = create_state(net, seed=1, dtype=torch.float32).to(device='cuda', dtype=torch.float32)
Z = torch.rand(50, 32, 32, 3).cuda(); X = torch.rand(256, 32, 32, 3).cuda()
for rep in range(3):
    acc.clear()
    torch.cuda.synchronize(); t = time.perf_counter()
    loss, g = TI.variational_grad_scalable(Z, X, st, 0.005, model_type="classifier", full_set_size=49000, x_chunk=128)
    torch.cuda.synchronize(); tot = time.perf_counter() - t
    print(f"rep {rep}: total {tot:.3f} s  " + "  ".join(f"{k} {v:.3f}" for k, v in acc.items()))
