#!/bin/bash
# On the GPU box: layer 3's launches with the column blocks INNER (dlip_debug_set(5, 1): round 4's order) vs PAIRED as lanes of
# the split (5, 2: round 5) -- time (interleaved, three rounds) and L2-miss traffic (rocprofv3 FETCH_SIZE x 2, WRITE_SIZE; separate
# passes as MI355X_MICROARCH.md prescribes).   tools/probes/paired_ab.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do for m in 1 2; do echo "== ninner $m round $i"; python3 $R/tools/bench_dma.py --only l3 --ninner $m --iters 20 2>&1 | grep -v amdgpu.ids; done; done
for m in 1 2; do
  for C in FETCH_SIZE WRITE_SIZE; do
    O=$R/gpurun_out/pmc_paired_${m}_$C; rm -rf $O; mkdir -p $O
    rocprofv3 --pmc $C --output-format csv -d $O -- python3 $R/tools/bench_dma.py --only l3 --ninner $m --iters 3 > $O/log.txt 2>&1
    python3 - "$O" "$m" "$C" <<'P'
import csv, glob, sys, collections
d, m, c = sys.argv[1:4]
agg = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_igemm_f16x3_dma" in r["Kernel_Name"] and r["Counter_Name"] == c:
            agg[(r["Kernel_Name"][40:100], r["Grid_Size"])].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    mult = 2 if c == "FETCH_SIZE" else 1     # gfx950: FETCH_SIZE reports half of a wide streaming read; unit KiB
    print(f"ninner {m} {c} {k}: launches {len(v)}, mean {sum(v) / len(v) * 1024 * mult / 1e6:.1f} MB per launch")
P
  done
done
