#!/bin/bash
# two ranks on the one GPU of a gpurun box, gloo instead of RCCL: exercises bench.py's N>1 path end to end.
# bench.py starts its own ranks (no torch.distributed.run): `--gpus 2` with no WORLD_SIZE in the environment.
export LIP_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
unset WORLD_SIZE RANK LOCAL_RANK
timeout -k 10 400 python bench.py --gpus 2 --steps 2 --warmup 1 --probes 128 --samples 0 --no-cpu-baseline --no-resnet50
# the same two ranks under strong scaling: ONE data set of 200 examples, 100 per rank as two chunks of 50
timeout -k 10 400 python bench.py --gpus 2 --steps 2 --warmup 1 --probes 64 --scaling strong --n-total 200 --no-resnet50
