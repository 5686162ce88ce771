#!/bin/bash
# A/B two library builds on one box: interleaved rounds of tools/bench_layers.py
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
  for L in A B; do
    if [ $L = A ]; then export DLIP_LIB_PATH=$R/deeplip_amd/lib/libdeeplip_hip_A.so; else export DLIP_LIB_PATH=$R/deeplip_amd/lib/libdeeplip_hip.so; fi
    echo "== lib $L round $i"; python3 $R/tools/bench_layers.py --iters 10 "$@" | tail -n +2
  done
done
