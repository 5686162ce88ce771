#!/bin/bash
# Build a variant of the library with extra compiler flags (experiments / ablations; the product library is
# untouched):  tools/variant.sh NAME -DFLAG ...   ->  deeplip_amd/lib/variants/libdeeplip_hip_NAME.so
# On the GPU box:  DLIP_LIB_PATH=deeplip_amd/lib/variants/libdeeplip_hip_NAME.so python tools/bench_dma.py ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
O=$R/deeplip_amd/lib/variants/$N.tmp
mkdir -p $O
for f in $R/deeplip_amd/csrc/*.hip; do
  s=$(basename $f .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -I$R/include -I$R/deeplip_amd/csrc -c $f -o $O/$s.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/deeplip_amd/lib/variants/libdeeplip_hip_$N.so $O/*.o
rm -rf $O
echo $R/deeplip_amd/lib/variants/libdeeplip_hip_$N.so
