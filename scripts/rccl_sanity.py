"""RCCL smoke on a one-GPU box: world_size 1 process group on backend "nccl" (= RCCL), the collectives bench.py and
lip_amd.dist issue (all_reduce sync + async on slices, all_gather, barrier)."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
Y = torch.arange(64 * 1000, device=dev, dtype=torch.float32).reshape(64, 1000)
ref = Y.clone()
hs = [dist.all_reduce(Y[c:c + 16], op=dist.ReduceOp.SUM, async_op=True) for c in range(0, 64, 16)]
for h in hs: h.wait()
dist.all_reduce(Y); dist.barrier(); torch.cuda.synchronize()
parts = [torch.empty_like(Y)]; dist.all_gather(parts, Y)
print("rccl ok", torch.equal(Y, ref), torch.equal(parts[0], ref), torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
dist.destroy_process_group()
