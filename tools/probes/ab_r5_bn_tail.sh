#!/bin/bash
# Same-box A/B (round 5): the training steps with ABI 43's BatchNorm launch sequence (dlip_debug_set(8, 0)) / every rows-kernel tile at
# full height (9, 0) against the built-in choices, interleaved; then C3 (speech encoder, B = 256) with and without the short last round.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for i in 1 2 3; do
  for v in "--dbg 8=0 --dbg 9=0" "--dbg 9=0" ""; do
    python3 $R/tools/bench_train_video.py --batch 32 --steps 10 $v 2>&1 | grep replayed | sed -E "s/.*: ([0-9.]+ ms\/step = [0-9.]+ clips\/s).*/video [$v]: \1/"
    python3 $R/tools/bench_train_audio.py --batch 256 --steps 10 $v 2>&1 | tail -1 | sed -E "s/.*: ([0-9.]+ ms\/step = [0-9.]+ utt\/s).*/audio [$v]: \1/"
  done
done
for i in 1 2; do
  for v in "--dbg 9=0" ""; do
    python3 $R/bench.py --no-cpu-baseline --single-mode $v 2>/dev/null | python3 -c "
import json, sys
b = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = b['configs']['C3_audio_embed']
print('C3 [$v]:', c['value'], c['ms_per_step'], c['step_frac'], c['dominant_kernel'], 'headline', b['value'], b['roofline']['frac'], 'F2', b['configs']['F2_train_video_step'].get('value'), b['configs']['F2_train_audio_step'].get('value'))"
  done
done
