"""torch.profiler view of one ResNet-50 stochastic inducing-point gradient step: which device ops the 6 s outside the
network sweeps go to."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lip_amd
from lip_amd.scalemodels import ResNet50
from lip_amd.toymodels import create_state
from lip_amd.train_inducing import variational_grad_stochastic
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
state = create_state(ResNet50(1000), seed=1, dtype=torch.float32).to(device=dev, dtype=torch.float32)
g = torch.Generator().manual_seed(6)
Z = torch.rand(2, 224, 224, 3, generator=g).to(dev); X = torch.rand(8, 224, 224, 3, generator=g).to(dev)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    variational_grad_stochastic(Z, X, state, 1.0, key=3, model_type="classifier", full_set_size=10000, st_samples=48, slq_samples=2, slq_num_matvecs=4)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=60))
print(prof.key_averages().table(sort_by="cpu_time_total", row_limit=12, max_name_column_width=60))
