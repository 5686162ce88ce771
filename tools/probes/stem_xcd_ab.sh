#!/bin/bash
# On the GPU box: the fused stem + pool kernel with frames dealt out in blockIdx order (library A: round 4) vs an XCD taking a
# contiguous block of frames (library B: round 5) -- time, interleaved, and L2-miss traffic (FETCH_SIZE x 2 / WRITE_SIZE, separate passes).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do for L in A B; do
  if [ $L = A ]; then export DLIP_LIB_PATH=$R/deeplip_amd/lib/libdeeplip_hip_A.so; else export DLIP_LIB_PATH=$R/deeplip_amd/lib/libdeeplip_hip.so; fi
  echo "== lib $L round $i"; python3 $R/tools/bench_stem.py 2>&1 | grep -v amdgpu.ids | tail -1
done; done
for L in A B; do
  if [ $L = A ]; then export DLIP_LIB_PATH=$R/deeplip_amd/lib/libdeeplip_hip_A.so; else export DLIP_LIB_PATH=$R/deeplip_amd/lib/libdeeplip_hip.so; fi
  for C in FETCH_SIZE WRITE_SIZE; do
    O=$R/gpurun_out/pmc_stem_${L}_$C; rm -rf $O; mkdir -p $O
    rocprofv3 --pmc $C --output-format csv -d $O -- python3 $R/tools/bench_stem.py > $O/log.txt 2>&1
    python3 - "$O" "$L" "$C" <<'P'
import csv, glob, sys, collections
d, m, c = sys.argv[1:4]
agg = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "stem" in r["Kernel_Name"] and r["Counter_Name"] == c:
            agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    mult = 2 if c == "FETCH_SIZE" else 1
    print(f"lib {m} {c} {k}: launches {len(v)}, mean {sum(v) / len(v) * 1024 * mult / 1e6:.1f} MB per launch")
P
  done
done
