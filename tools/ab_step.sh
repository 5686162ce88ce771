#!/bin/bash
# A/B two library builds on the WHOLE STEP on one box (bench.py, plan replay): at the board's power limit a kernel change that
# wins when a layer runs back to back (tools/bench_dma.py) can lose in the step -- this is the A/B that decides.
#   here:  tools/ab.sh snapshot <commit>   -> deeplip_amd/lib/libdeeplip_hip_A.so
#   box:   tools/ab_step.sh [rounds]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for i in $(seq 1 ${1:-2}); do
  for L in A B; do
    if [ $L = A ]; then export DLIP_LIB_PATH=$R/deeplip_amd/lib/libdeeplip_hip_A.so; else export DLIP_LIB_PATH=$R/deeplip_amd/lib/libdeeplip_hip.so; fi
    python3 $R/bench.py --no-cpu-baseline --no-configs --single-mode 2>/dev/null | python3 -c "
import json, sys
b = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = b['roofline']; k = r['kernels']
print('$L', b['value'], b['ms_per_step'], 'sum', r['kernels_ms_sum'], ' '.join(f\"{n.split('kernel')[-1]}:{v['tflops']:.0f}\" for n, v in k.items() if v['tflops'] > 50))"
  done
done
