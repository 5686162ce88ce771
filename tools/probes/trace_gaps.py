#!/usr/bin/env python3
"""Gaps between consecutive kernels of the replayed step, per HIP stream (queue), from a rocprofv3 --kernel-trace CSV
(development probe): how much of a step is the chip waiting between launches of one stream?
    rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 bench.py --steps 20 --no-cpu-baseline --no-configs --single-mode --no-kernel-events
    python3 tools/probes/trace_gaps.py /tmp/tr"""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for q, ks in byq.items():
    if len(ks) < 200:
        continue
    ks = ks[len(ks) // 2:]                      # steady state: the second half of the run
    gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
    small = [g for g in gaps if 0 <= g < 50000]
    busy = sum(e - s for s, e, _ in ks)
    span = ks[-1][1] - ks[0][0]
    small.sort()
    print(f"queue {q}: {len(ks)} kernels, busy {busy / 1e6:.2f} ms of {span / 1e6:.2f} ms; gaps < 50 us: n={len(small)} sum {sum(small) / 1e6:.3f} ms "
          f"median {small[len(small) // 2] / 1e3:.2f} us p90 {small[int(len(small) * 0.9)] / 1e3:.2f} us; negative (overlapping) gaps: {sum(1 for g in gaps if g < 0)}")
