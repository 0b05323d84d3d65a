#!/bin/bash
# two ranks on the one GPU of a gpurun box, gloo instead of RCCL: exercises bench.py's N>1 path end to end
export LIP_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 2 --steps 2 --warmup 1 --probes 128 --samples 50 --no-cpu-baseline --no-resnet50
