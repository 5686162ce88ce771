#!/bin/bash
# On the GPU box: HBM bytes of one bench_dma.py layer with the balanced split (stream-K slabs) and without it (plain launch: no
# slabs) -> what part of the dominant kernel's traffic is slab hand-off and what part is operands.
#   tools/pmc_traffic_split.sh "l3.conv l4.conv tdnn.k3d2"
R=${GRAFT_REPO_ROOT:-/root/repo}
LAYERS=${1:-l3.conv}
cd /tmp && export TMPDIR=/tmp
for L in $LAYERS; do
 for SK in ${SKS:--1 0}; do   # SKS="-1 3": the L2-local hand-off experiment (dlip_debug_set(3, 3))
  for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    N=$(echo $C | cut -d" " -f1)
    O=$R/gpurun_out/tsplit_${L}_sk${SK}_$N
    rm -rf $O; mkdir -p $O
    rocprofv3 --pmc $C --output-format csv -d $O -- python3 $R/tools/bench_dma.py --only $L --iters 3 --streamk $SK > $O/log.txt 2>&1
  done
  python3 - "$R/gpurun_out" "$L" "$SK" <<'P'
import csv, glob, sys, collections
root, L, SK = sys.argv[1:4]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for N in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum"):
    for f in glob.glob(f"{root}/tsplit_{L}_sk{SK}_{N}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "conv_" not in k: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k, c in agg.items():
    out = {x: y / len(disp[(k, x)]) for x, y in c.items()}
    rd = 2 * out.get("FETCH_SIZE", 0) * 1024 / 1e6     # KiB units; x2: gfx950 FETCH_SIZE correction (MI355X_MICROARCH.md)
    wr = out.get("WRITE_SIZE", 0) * 1024 / 1e6
    print(f"{L} streamk={SK} {k[40:100]}: read {rd:.1f} MB write {wr:.1f} MB per launch;  L2 hit {out.get('TCC_HIT_sum',0):.3g} miss {out.get('TCC_MISS_sum',0):.3g}")
P
 done
done
