#!/bin/bash
# FETCH_SIZE per kernel for the default and the alt build (one box call)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for v in default alt; do
  if [ $v = alt ]; then export LIP_LIB_PATH=$R/laplace-inducing-points_amd/csrc/alt/liblip_hip.so; else unset LIP_LIB_PATH; fi
  rm -rf $R/gpurun_out/trf_$v
  rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/trf_$v -o run -- python3 $R/bench.py --steps 2 --warmup 1 --samples 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/trf_$v.err
  python3 - <<PY
import csv, collections
acc=collections.defaultdict(lambda:[0.0,0,0.0])
for r in csv.DictReader(open("$R/gpurun_out/trf_$v/run_counter_collection.csv")):
    k=r["Kernel_Name"].split("(")[0].replace("void lip::","")
    if "igemm_fast" not in k: continue
    a=acc[k]; a[0]+=float(r["Counter_Value"]); a[1]+=1; a[2]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))*1e-6
for k,a in acc.items():
    print("$v", k[:48], "read GB/launch", round(2*a[0]*1024/a[1]/1e9,2), "ms", round(a[2]/a[1],3))
PY
done
