import sys, torch, numpy as np
sys.path.insert(0, '.')
import lip_amd
from lip_amd.engine import LinearizedNet
from lip_amd.scalemodels import ResNet50
from lip_amd.toymodels import create_state
net = ResNet50(1000); st = create_state(net, seed=1, dtype=torch.float32)
Z = torch.rand(1, 224, 224, 3, generator=torch.Generator().manual_seed(3))
eng = LinearizedNet(st, Z.cuda(), "classifier", workspace_bytes=4 << 30, max_chunk=2)
torch.cuda.synchronize()
np.save(sys.argv[1], eng.prim.cpu().numpy())
m = eng.cn.meta
import json
json.dump({k: ({str(a): int(b) for a, b in v.items() if b is not None} if isinstance(v, dict) else None) for k, v in m.items() if k in ("dphi_off", "xhat_off", "amax_off")}, open(sys.argv[1] + ".json", "w"))
