"""One matrix-free gradient step of the stochastic inducing-point objective at BASELINE configs[4]: ResNet-50
(25.6 M parameters, K = 1000) at 224 x 224, M inducing images, a data batch of Kb images.  Prints stage timings.
Usage: python scripts/r50_inducing_grad.py [M] [Kb] [st_samples] [k_slq]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lip_amd  # noqa: E402,F401
from lip_amd.scalemodels import ResNet50  # noqa: E402
from lip_amd.toymodels import create_state  # noqa: E402
from lip_amd.train_inducing import variational_grad_stochastic  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 2
Kb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
st = int(sys.argv[3]) if len(sys.argv) > 3 else 48
k = int(sys.argv[4]) if len(sys.argv) > 4 else 4
dev = torch.device("cuda", 0)
state = create_state(ResNet50(1000), seed=1, dtype=torch.float32).to(device=dev, dtype=torch.float32)
g = torch.Generator().manual_seed(6)
Z = torch.rand(M, 224, 224, 3, generator=g).to(dev)
X = torch.rand(Kb, 224, 224, 3, generator=g).to(dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
val, gZ, info = variational_grad_stochastic(Z, X, state, 1.0, key=3, model_type="classifier", full_set_size=10000, st_samples=st,
                                            slq_samples=2, slq_num_matvecs=k, return_terms=True)
torch.cuda.synchronize()
t1 = time.perf_counter() - t0
print(f"M={M} Kb={Kb} st={st} k={k}: {t1:.2f} s  value {val:.6e}  directions {info['directions']}  grad finite {bool(torch.isfinite(gZ).all())} "
      f"|g| {gZ.norm().item():.3e}  peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
print("stages:", {k: round(v, 3) for k, v in info["stage_seconds"].items()})
