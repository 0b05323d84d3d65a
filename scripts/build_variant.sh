#!/bin/bash
# A/B build of liblip_hip.so with extra compile flags for lip_mfma.hip: build/ab/lib_<name>.so (select with LIP_LIB_PATH).
# usage: scripts/build_variant.sh <name> "<extra flags>" [nosched]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/laplace-inducing-points_amd/csrc
OUT=$ROOT/build/ab
mkdir -p $OUT
SCHED="-fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp"
[ "$3" = nosched ] && SCHED=""
make -C $CS -j4 >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I$CS -Wall -Wno-unused-function $SCHED $2 -c $CS/lip_mfma.hip -o $OUT/mfma_$1.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OUT/mfma_$1.o $CS/lip_small.o $CS/lip_krylov.o $CS/lip_engine.o -o $OUT/lib_$1.so
echo built $OUT/lib_$1.so
