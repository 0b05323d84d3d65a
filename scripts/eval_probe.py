"""One evaluation batch at the CIFAR config (B = 256 test images, S = 200 draws): the reference's sample-based predictive
(src/lla.py:133-156 through S tangent sweeps) against the closed-form per-point marginals."""
import sys, time, torch
sys.path.insert(0, '.')
import lip_amd
from lip_amd.scalemodels import ResNet1M
from lip_amd.toymodels import create_state
from lip_amd import lla

net = ResNet1M(10)
st = create_state(net, seed=1, dtype=torch.float32).to(device='cuda', dtype=torch.float32)
Z = torch.rand(50, 32, 32, 3).cuda(); X = torch.rand(256, 32, 32, 3).cuda()
for rep in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    s = lla.predict_lla_scalable(st, X, Z, "classifier", 0.005, key=rep, full_set_size=49000, num_samples=200)
    torch.cuda.synchronize(); t1 = time.perf_counter() - t
    t = time.perf_counter()
    d = lla.predict_lla_marginals(st, X, Z, "classifier", 0.005, full_set_size=49000, batch=64)
    m = d.sample((200,), seed=rep)
    torch.cuda.synchronize(); t2 = time.perf_counter() - t
    ps = torch.softmax(s, -1).mean(0); pm = torch.softmax(m.float(), -1).mean(0)
    print(f"rep {rep}: sampled {t1:.3f} s, marginals {t2:.3f} s; mean-prob diff {float((ps - pm).abs().max()):.4f}; var ratio {float(s.var(0).mean() / m.var(0).mean()):.3f}")
