"""Where the non-product time of the D-space Lanczos sampler goes (bench leg `lanczos_sampler`: 256 recurrences, k = 36
on the CIFAR binding): torch.profiler table of one call."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lip_amd
from lip_amd.sample import sample_lanczos
from lip_amd.scalemodels import ResNet1M
from lip_amd.toymodels import create_state
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
st = create_state(ResNet1M(10), seed=1, dtype=torch.float32).to(device=dev, dtype=torch.float32)
Z = torch.rand(50, 32, 32, 3, generator=torch.Generator().manual_seed(0)).to(dev)
D = 1084586
sample_lanczos(st, Z, D, 0.005, 5, "classifier", num_samples=8, full_set_size=49000, num_matvecs=4)
torch.cuda.synchronize()
t0 = time.perf_counter()
sample_lanczos(st, Z, D, 0.005, 6, "classifier", num_samples=256, full_set_size=49000, num_matvecs=36)
torch.cuda.synchronize()
print("seconds", time.perf_counter() - t0)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    sample_lanczos(st, Z, D, 0.005, 6, "classifier", num_samples=256, full_set_size=49000, num_matvecs=36)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=70))
