#!/bin/bash
# On the GPU box: PMC counters of the rows kernel on the k = 1 TDNN GEMM (B = 64, 296 frames, 512 -> 512) next to the ring
# kernel's 128x128 instance on the same launch -- VALU per MFMA, issue / wait shares, matrix-pipe busy.
#   tools/pmc_rows.sh > profiles/r4/pmc_k1_gemm.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
CTRS="SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
cd /tmp && export TMPDIR=/tmp
for MODE in 0 5; do
  O=/tmp/pmc_rows_$MODE; rm -rf $O; mkdir -p $O
  rocprofv3 --pmc $CTRS --output-format csv -d $O -- python3 $R/tools/probes/rows_one.py $MODE > $O/log.txt 2>&1
  python3 - "$O" "$MODE" <<'P'
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
    per = {x: y / nl for x, y in sorted(c.items())}
    gui = per["GRBM_GUI_ACTIVE"] / 8
    print(("rows kernel (dlip_debug_set 6 = 5)" if v == "5" else "ring kernel (rows kernel off)"), k[28:100], "launches", nl)
    print("   ", {x: round(y) for x, y in per.items()})
    print("    VALU per MFMA %.2f   matrix pipe busy %.3f   waves: issuing %.2f / issue-stalled %.2f / waiting %.2f of their cycles" % (
        per["SQ_INSTS_VALU"] / per["SQ_INSTS_MFMA"], per["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024),
        per["SQ_ACTIVE_INST_ANY"] / (per["SQ_ACTIVE_INST_ANY"] + per["SQ_WAIT_INST_ANY"] + per["SQ_WAIT_ANY"]),
        per["SQ_WAIT_INST_ANY"] / (per["SQ_ACTIVE_INST_ANY"] + per["SQ_WAIT_INST_ANY"] + per["SQ_WAIT_ANY"]),
        per["SQ_WAIT_ANY"] / (per["SQ_ACTIVE_INST_ANY"] + per["SQ_WAIT_INST_ANY"] + per["SQ_WAIT_ANY"])))
P
done
