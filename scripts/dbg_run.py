import sys, torch
sys.path.insert(0, '.')
import lip_amd
from lip_amd import krylov
from lip_amd.engine import LinearizedNet
from lip_amd.scalemodels import ResNet1M
from lip_amd.toymodels import create_state
net = ResNet1M(10); st = create_state(net, seed=1, dtype=torch.float32)
eng = LinearizedNet(st, torch.rand(50, 32, 32, 3).cuda(), "classifier", workspace_bytes=24 << 30, max_chunk=256)
V = krylov.fill_rademacher(256, eng.D, 1, "cuda")
eng.ggn_vp(V, 1.0, 0.0); torch.cuda.synchronize()
