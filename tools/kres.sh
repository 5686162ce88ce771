#!/bin/bash
# Per-kernel register / scratch / occupancy table for one .hip source (cross-compiles, no GPU needed).
src=$1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$(dirname "$0")/../include" -I"$(dirname "$0")/../deeplip_amd/csrc" \
  -c "$src" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size" \
  | sed -E 's/.*remark: +//; s/ \[-Rpass.*//; s/Function Name: _ZN12_GLOBAL__N_1[0-9]*//' | paste - - - - -
