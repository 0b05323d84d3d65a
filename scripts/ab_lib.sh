#!/bin/bash
# A/B of two builds of the library in ONE box call: default vs laplace-inducing-points_amd/csrc/alt/liblip_hip.so
for rep in 1 2; do
for v in default alt; do
  if [ $v = alt ]; then export LIP_LIB_PATH=$PWD/laplace-inducing-points_amd/csrc/alt/liblip_hip.so; else unset LIP_LIB_PATH; fi
  timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-resnet50 --samples 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', round(d['value'],1), {k:round(v['tflops'],1) for k,v in d['roofline']['per_kernel'].items()})"
done
done
