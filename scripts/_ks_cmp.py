import sys, numpy as np, torch
sys.path.insert(0, '.')
import lip_amd
from lip_amd.scalemodels import ResNet50
from lip_amd.toymodels import create_state
a = np.load('/tmp/y_ks.npy'); b = np.load('/tmp/y_noks.npy')
scale = np.abs(b).max()
print("max rel", np.abs(a - b).max() / scale)
net = ResNet50(1000); st = create_state(net, seed=1, dtype=torch.float32)
# walk leaves in flat order
from lip_amd.utils import flatten_nn_params
flat, unravel = flatten_nn_params(st.params)
off = 0
def walk(t, pre=""):
    global off
    if isinstance(t, dict):
        for k in t: walk(t[k], pre + "/" + str(k))
    else:
        n = t.numel(); e = np.abs(a[off:off+n] - b[off:off+n]).max() / scale
        if e > 1e-6: print(f"{pre:60s} off {off:9d} n {n:8d} shape {tuple(t.shape)} err {e:.2e}")
        off += n
walk(unravel(flat))
print("D", off)
