#!/bin/bash
# Diagnostic build of the library with in-kernel s_memtime stamps in the LDS-DMA conv kernel (a separate
# .so; the product library is untouched).  Build here, then on the GPU box:
#   DLIP_LIB_PATH=deeplip_amd/lib/stamps/libdeeplip_hip_stamps.so DLIP_STAMP_PRINT=1 python tools/bench_dma.py --iters 1
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/deeplip_amd/lib/stamps
mkdir -p $O
for f in $R/deeplip_amd/csrc/*.hip; do
  s=$(basename $f .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDLIP_STAMPS -I$R/include -I$R/deeplip_amd/csrc -c $f -o $O/$s.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libdeeplip_hip_stamps.so $O/*.o
rm -f $O/*.o
echo $O/libdeeplip_hip_stamps.so
