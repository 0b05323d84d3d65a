#!/bin/bash
# A/B of an environment switch in ONE box call:  bash scripts/ab_env.sh LIP_NOBV4
VAR=$1
for rep in 1 2; do
for v in 0 1; do
  if [ $v = 1 ]; then export $VAR=1; else unset $VAR; fi
  timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-resnet50 --samples 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$v', round(d['value'],1), {k:round(v['tflops'],1) for k,v in d['roofline']['per_kernel'].items()})"
done
done
