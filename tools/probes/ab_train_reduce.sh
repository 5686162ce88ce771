#!/bin/bash
# Same-box A/B of the training steps: in-kernel split finisher (dlip_debug_set(3, 4)) vs the reduce launch (default), interleaved.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for m in 4 -1 4 -1; do
  echo "== dlip_debug_set(3, $m)   (4 = in-kernel finisher, -1 = built-in: reduce launch for few tiles cut many ways)"
  python3 $R/tools/bench_train_video.py --batch 32 --steps 10 --dbg 3=$m 2>&1 | grep "replayed" | cut -c1-170
  python3 $R/tools/bench_train_audio.py --batch 256 --steps 10 --dbg 3=$m 2>&1 | tail -1 | cut -c1-170
done
