#!/bin/bash
# On the GPU box: time tools/bench_dma.py with the product library and each named variant library
# (tools/variant.sh), two interleaved rounds.  usage: tools/abl.sh "v1 v2 ..." [bench_dma.py args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V=$1; shift
for i in 1 2; do
  for L in base $V; do
    if [ $L = base ]; then unset DLIP_LIB_PATH; else export DLIP_LIB_PATH=$R/deeplip_amd/lib/variants/libdeeplip_hip_$L.so; fi
    echo "== lib $L round $i"; timeout -k 10 200 python3 $R/tools/bench_dma.py --iters 10 "$@" 2>&1 | grep -v amdgpu.ids
  done
done
