#!/bin/bash
# A/B two library builds on one box (device-to-device variance is ~10 %: only same-box numbers compare).
#   here:  tools/ab.sh snapshot      -> builds the committed HEAD sources into deeplip_amd/lib/libdeeplip_hip_A.so
#   box:   tools/ab.sh run [bench_dma.py args]   -> interleaved rounds, A = snapshot, B = working tree build
# (the ring kernel's lock-step loop, round 3's `nopp` A/B, is tile 11 of the lab build now: python -m deeplip_amd.build --lab)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
if [ "$1" = snapshot ]; then
  T=$(mktemp -d); git -C $R archive ${2:-HEAD} deeplip_amd/csrc include | tar -x -C $T
  for f in $T/deeplip_amd/csrc/*.hip; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$T/include -I$T/deeplip_amd/csrc -c $f -o $T/$(basename $f .hip).o &
  done; wait
  echo 'extern "C" const char* dlip_source_sha(void) { return "snapshot"; }' > $T/stamp.cpp
  /opt/rocm/bin/hipcc -O2 -fPIC -c -x c++ $T/stamp.cpp -o $T/stamp.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/deeplip_amd/lib/libdeeplip_hip_A.so $T/*.o && rm -rf $T
  echo built $R/deeplip_amd/lib/libdeeplip_hip_A.so from HEAD; exit 0
fi
shift
for i in 1 2; do
  for L in A B; do
    if [ $L = A ]; then export DLIP_LIB_PATH=$R/deeplip_amd/lib/libdeeplip_hip_A.so; else export DLIP_LIB_PATH=$R/deeplip_amd/lib/libdeeplip_hip.so; fi
    echo "== lib $L round $i"; python3 $R/tools/bench_dma.py --iters 10 "$@" 2>&1 | grep -v amdgpu.ids
  done
done
