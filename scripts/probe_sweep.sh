#!/bin/bash
# throughput vs probe-block size (the N > 1 path sweeps chunks of the block; Krylov loops on one vector run P = 1)
for P in ${@:-1 2 4 8 16 32 64 128 256}; do
  timeout -k 10 300 python bench.py --steps 4 --warmup 2 --probes $P --no-cpu-baseline --no-resnet50 --samples 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('P=$P', round(d['value'],1), 'GGN-vp/s', round(d['ms_per_step'],2), 'ms/step', {k:round(v['tflops'],1) for k,v in d['roofline']['per_kernel'].items()})"
done
