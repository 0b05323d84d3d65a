import sys, os, torch, numpy as np
sys.path.insert(0, '.')
import lip_amd
from lip_amd.engine import LinearizedNet
from lip_amd.scalemodels import ResNet50
from lip_amd.toymodels import create_state
from lip_amd.utils import flatten_nn_params
torch.manual_seed(0)
net = ResNet50(1000); st = create_state(net, seed=1, dtype=torch.float32)
Z = torch.rand(1, 224, 224, 3, generator=torch.Generator().manual_seed(3))
eng = LinearizedNet(st, Z.cuda(), "classifier", workspace_bytes=4 << 30, max_chunk=2)
v = torch.randn(eng.D, generator=torch.Generator().manual_seed(4))
out = eng.ggn_vp(v.cuda()[None], 1.0, 0.0)[0].cpu().numpy()
np.save(sys.argv[1], out)
