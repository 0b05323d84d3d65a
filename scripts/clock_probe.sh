#!/bin/bash
# samples shader clock / power while the headline sweep runs (rocm-smi as an ordinary user)
python bench.py --steps 60 --warmup 5 --samples 0 --no-cpu-baseline > gpurun_out/clock_bench.json 2>/dev/null &
BP=$!
sleep 6
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power" | head -6
  echo ---
  sleep 1
done
wait $BP
python -c "
import json; d=json.load(open('gpurun_out/clock_bench.json')); print('value', d['value'])"
