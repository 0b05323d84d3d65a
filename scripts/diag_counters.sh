#!/bin/bash
# diagnostic PMC passes (separate runs; counters only with --kernel-trace): which unit the MFMA kernels wait on
# (SQ counters only: a pass of TA_*/TCP_* derived sums did not finish within 7 minutes on the pool and was dropped)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/diag; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
ARGS="$R/bench.py --steps 2 --warmup 1 --samples 0 --no-cpu-baseline"
i=0
while read -r set; do
  i=$((i+1))
  rocprofv3 --output-format csv --kernel-trace --pmc $set -d "$OUT/p$i" -o run -- python3 $ARGS > /dev/null 2> "$OUT/p$i.err" || echo "pass $i failed"
  echo "pass $i done: $set"
done <<'SETS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAVES
SETS
find "$OUT" -name "*.csv" ! -name "*_counter_collection.csv" -delete
du -sh "$OUT"
