#!/bin/bash
# throughput vs probe-block size (the N > 1 path sweeps chunks of P/4 so that the all-reduce of one hides behind the next)
for P in 32 64 128 256; do
  timeout -k 10 300 python bench.py --steps 4 --warmup 2 --probes $P --no-cpu-baseline --no-resnet50 --samples 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('P=$P', round(d['value'],1), round(d['ms_per_step'],2))"
done
