#!/usr/bin/env python3
"""ONE replayed optimisation step of the lip-clip trainer from a rocprofv3 kernel trace: the launches between the last two stem
launches of the run (the stem runs once per step), totals per kernel name.
    rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 tools/bench_train_video.py --batch 32 --steps 6
    python3 tools/probes/train_step_kernels.py /tmp/tr [marker-kernel-substring]"""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mark = sys.argv[2] if len(sys.argv) > 2 else "stem3d_f16x3_kernel"
idx = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"] and "split_stem" not in r["Kernel_Name"]]
if len(idx) < 3:
    sys.exit(f"fewer than three '{mark}' launches in the trace")
a, b = idx[-2], idx[-1]
step = rows[a:b]
wall = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e6
agg = collections.OrderedDict()
for r in step:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "", 1).split("(")[0]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    e = agg.setdefault(n, [0, 0.0])
    e[0] += 1; e[1] += d
tot = sum(v[1] for v in agg.values()) / 1e3
# how much of the step's wall time has at least one / more than one kernel on the chip, and how much of the kernel time is launches under 10 us
ev = []
for r in step:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
busy1 = busy2 = 0; depth = 0; last = ev[0][0]
for t, d in ev:
    if depth >= 1: busy1 += t - last
    if depth >= 2: busy2 += t - last
    depth += d; last = t
# time a kernel had the chip to itself (depth 1), by kernel name: what the step's wall time is actually made of
solo = collections.defaultdict(float)
ev2 = []
for k, r in enumerate(step):
    ev2.append((int(r["Start_Timestamp"]), 1, k)); ev2.append((int(r["End_Timestamp"]), -1, k))
ev2.sort()
running = set(); last = ev2[0][0]
for t, d, k in ev2:
    if len(running) == 1:
        n = step[next(iter(running))]["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "", 1).split("(")[0]
        solo[n] += (t - last) / 1e3
    if d > 0: running.add(k)
    else: running.discard(k)
    last = t
small = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in step]
print(f"# chip busy (>= 1 kernel) {busy1 / 1e6:.2f} ms of the wall, >= 2 kernels at once {busy2 / 1e6:.2f} ms, idle {wall - busy1 / 1e6:.2f} ms; "
      f"{sum(1 for d in small if d < 10)} launches under 10 us = {sum(d for d in small if d < 10) / 1e3:.2f} ms of kernel time")
print(f"# one replayed training step, rocprofv3 kernel trace: wall {wall:.2f} ms, sum of kernel durations {tot:.2f} ms, {len(step)} launches")
print(f"# {'kernel':108s} {'calls':>5s} {'ms':>8s} {'avg us':>8s} {'alone ms':>9s}")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n[:110]:110s} {c:5d} {t / 1e3:8.3f} {t / c:8.1f} {solo.get(n, 0.0) / 1e3:9.3f}")
