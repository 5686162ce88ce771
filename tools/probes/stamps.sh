#!/bin/bash
# In-kernel s_memtime stamps of the LDS-DMA conv kernel: they exist in the LAB build only
# (python -m deeplip_amd.build --lab -> deeplip_amd/lib/libdeeplip_hip_lab.so; the product library has none).
# Build here, then on the GPU box:
#   DLIP_LIB_PATH=deeplip_amd/lib/libdeeplip_hip_lab.so DLIP_STAMP_PRINT=1 python3 tools/bench_dma.py --iters 1 --only l1.conv
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R && python3 -m deeplip_amd.build --lab
