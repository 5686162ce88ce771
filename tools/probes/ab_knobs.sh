#!/bin/bash
# Whole-step A/B of launch-rule knobs (bench.py --dbg KEY=VALUE -> dlip_debug_set) on one box, two rounds, interleaved.
#   tools/probes/ab_knobs.sh "" "--dbg 1=5" "--dbg 3=2" ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
run() { python3 $R/bench.py --no-cpu-baseline --no-configs --single-mode $1 2>/dev/null | python3 -c "
import json, sys
b = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = b['roofline']; k = r['kernels']
print('$1'.ljust(14), b['value'], b['ms_per_step'], 'sum', r['kernels_ms_sum'], ' '.join(f\"{n.split('kernel')[-1]}:{v['ms_per_step']:.3f}\" for n, v in k.items() if v['ms_per_step'] > 0.05))"; }
for i in 1 2; do for a in "$@"; do run "$a"; done; done
