"""Headline product in both precision modes on the bench binding: GGN-vp/s, split-vs-f32 difference, run-to-run
reproducibility.  Usage: python scripts/ab_precision.py [P]   (select a build with LIP_LIB_PATH)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lip_amd  # noqa: E402,F401
from lip_amd import krylov  # noqa: E402
from lip_amd.engine import LinearizedNet, set_precision  # noqa: E402
from lip_amd.scalemodels import ResNet1M  # noqa: E402
from lip_amd.toymodels import create_state  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
state = create_state(ResNet1M(10), seed=1231231234, dtype=torch.float32)
Z = torch.rand(50, 32, 32, 3, generator=torch.Generator().manual_seed(280300))
eng = LinearizedNet(state, Z.to(dev), "classifier", device=dev, workspace_bytes=24 << 30, max_chunk=P)
V = krylov.fill_rademacher(P, eng.D, 1234, dev)
out = {}
for mode in ("f32", "bf16x3"):
    set_precision(mode)
    Y1 = eng.ggn_vp(V, 980.0, 0.005).clone()
    Y2 = eng.ggn_vp(V, 980.0, 0.005).clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        eng.ggn_vp(V, 980.0, 0.005)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    out[mode] = (Y1, P / dt, float(((Y1 - Y2).abs().max() / Y1.abs().max()).item()))
set_precision("f32")
d = float(((out["bf16x3"][0] - out["f32"][0]).abs().max() / out["f32"][0].abs().max()).item())
print(f"P={P} f32 {out['f32'][1]:.0f} GGN-vp/s (run-to-run {out['f32'][2]:.2e}); bf16x3 {out['bf16x3'][1]:.0f} GGN-vp/s "
      f"(run-to-run {out['bf16x3'][2]:.2e}); split vs f32 {d:.3e}")
