#!/bin/bash
# Same-box A/B of the two training steps between the snapshot library (tools/ab.sh snapshot <commit> -> libdeeplip_hip_A.so) and
# the working tree's build, interleaved.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for i in 1 2; do
  for L in A B; do
    if [ $L = A ]; then export DLIP_LIB_PATH=$R/deeplip_amd/lib/libdeeplip_hip_A.so; else unset DLIP_LIB_PATH; fi
    python3 $R/tools/bench_train_video.py --batch 32 --steps 10 2>&1 | grep replayed | cut -c80-125 | sed "s/^/$L video: /"
    python3 $R/tools/bench_train_audio.py --batch 256 --steps 10 2>&1 | tail -1 | cut -c60-110 | sed "s/^/$L audio: /"
  done
done
