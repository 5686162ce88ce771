#!/bin/bash
# On the GPU box: PMC counters of one bench_dma.py layer for a list of tile variants.
#   tools/pmc_layer.sh l3.conv "5 19" [counters...]
R=${GRAFT_REPO_ROOT:-/root/repo}
LAYER=$1; VARS=$2; shift 2
CTRS=${@:-SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY}
cd /tmp && export TMPDIR=/tmp
for v in $VARS; do
  O=$R/gpurun_out/pmc_${LAYER}_v$v
  rm -rf $O; mkdir -p $O
  rocprofv3 --pmc $CTRS --output-format csv -d $O -- python3 $R/tools/bench_dma.py --only $LAYER --variants $v --iters 3 > $O/log.txt 2>&1
  python3 - "$O" "$v" <<'P'
import csv, glob, sys, collections
d, v = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv_" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, c in agg.items():
    nl = max(n[(k, x)] for x in c)
    print("v" + v, k[40:110], "launches", nl)
    print("   ", {x: round(y / nl) for x, y in sorted(c.items())})
P
done
