"""float32 engine vs float64 tape emulator on the sine regressor (generic kernels): jvp / vjp / ggn_vp relative errors
and the CG residual of test_nullproj — used to tell a rounding-order change from a defect between two builds."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lip_amd  # noqa
from lip_amd import _native as nv, krylov
from lip_amd.engine import LinearizedNet, build_consts
from lip_amd.toymodels import SimpleRegressor, create_state
from lip_amd.utils import flatten_nn_params
from tape_emulator import TapeMachine
import numpy as np
F64 = torch.float64
d = np.load(os.path.join(ROOT, "tests", "golden", "sine.npz"))
X = torch.tensor(d[d.files[0]], dtype=F64).reshape(-1, 1)[:16]
st = create_state(SimpleRegressor(16, 2) if False else SimpleRegressor(8, 4), 3, dtype=F64, logvar=-0.3)
eng = LinearizedNet(st, X, "regressor", workspace_bytes=1 << 26, max_chunk=4)
flat, _ = flatten_nn_params(st.params)
tm = TapeMachine(eng.cn, flat, build_consts(eng.cn, st.params, st.batch_stats, "cpu", F64), X, chunk=eng.chunk)
tm.primal()
g = torch.Generator().manual_seed(1)
V = torch.randn(4, eng.D, dtype=F64, generator=g) * 10
U = torch.randn(4, eng.n, eng.K, dtype=F64, generator=g)
rel = lambda a, b: ((a.double().cpu() - b).abs().max() / b.abs().max()).item()
print("D", eng.D, "jvp", rel(eng.jvp(V, "lt", 1.3), tm.jvp(V, nv.HEAD_LT, 1.3)), "vjp", rel(eng.vjp(U, "l", 0.7), tm.vjp(U, nv.HEAD_L, 0.7)),
      "ggn", rel(eng.ggn_vp(V, 2.0, 0.1), tm.ggn_vp(V, 2.0, 0.1)))
