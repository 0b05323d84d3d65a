#!/bin/bash
# rocprofv3 evidence for profiles/: one --stats pass and three separate --pmc passes of the same bench command
# (counters never combined with other trace domains), one --stats pass of the FULL bench (sampler, Lanczos, trace and
# Krylov legs) and stats + traffic passes of the Krylov / tall-skinny kernels alone.
# Run on the GPU box:  bash scripts/profile.sh r2
# then here:  python scripts/summarize_profile.py gpurun_out/prof_r2 r2
#             python scripts/summarize_profile.py gpurun_out/prof_r2/krylov r2_krylov "python scripts/krylov_bench.py"
#             cp gpurun_out/prof_r2/full/*/*_kernel_stats.csv profiles/r2_full_kernel_stats.csv
set -e
TAG=${1:-r2}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
ARGS="$R/bench.py --steps 5 --warmup 2 --samples 0 --no-cpu-baseline"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats" -o run -- python3 $ARGS > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
echo "stats pass done"
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o run -- python3 $ARGS > /dev/null 2> "$OUT/fetch.err"
echo "fetch pass done"
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d "$OUT/write" -o run -- python3 $ARGS > /dev/null 2> "$OUT/write.err"
echo "write pass done"
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d "$OUT/sq" -o run -- python3 $ARGS > /dev/null 2> "$OUT/sq.err"
echo "sq pass done"
FULL="$R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-resnet50"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/full" -o run -- python3 $FULL > "$OUT/bench_full_under_rocprof.json" 2> "$OUT/full.err"
echo "full stats pass done"
K="$R/scripts/krylov_bench.py"
cd "$R"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/krylov/stats" -o run -- python3 $K > "$OUT/krylov_bench_under_rocprof.json" 2> "$OUT/krylov_stats.err"
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d "$OUT/krylov/fetch" -o run -- python3 $K > /dev/null 2> "$OUT/krylov_fetch.err"
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d "$OUT/krylov/write" -o run -- python3 $K > /dev/null 2> "$OUT/krylov_write.err"
echo "krylov passes done"
# keep only what the summariser reads (the merge back is capped at 64 MiB)
find "$OUT" -name "*.csv" ! -name "*_kernel_stats.csv" ! -name "*_counter_collection.csv" -delete
du -sh "$OUT"
