#!/usr/bin/env python3
"""One replayed step of bench.py's timed region as a per-queue timeline (rocprofv3 --kernel-trace CSV): every kernel with start
offset, duration and the gap to its predecessor on the same queue -- what lies between the MFMA launches' spans and ms_per_step.
    rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 bench.py --steps 20 --no-cpu-baseline --no-configs --single-mode --no-kernel-events
    python3 tools/probes/step_timeline.py /tmp/tr"""
import csv, glob, sys, collections, re
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step ends with the span collect kernel (or the range verdict kernel): take the step in the middle of the run
marks = [i for i, r in enumerate(rows) if "znorm_cat" in r["Kernel_Name"]]
if len(marks) < 8:
    sys.exit("no steps found")
a, b = marks[len(marks) // 2], marks[len(marks) // 2 + 1]
step = rows[a + 1:b + 1]
t0 = min(int(r["Start_Timestamp"]) for r in step)
t1 = max(int(r["End_Timestamp"]) for r in step)
print(f"step: {len(step)} kernels, {(t1 - t0) / 1e3:.1f} us from first start to last end")
last = {}
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void |\(anonymous namespace\)::", "", n))[:70]
tot = collections.defaultdict(float)
for r in step:
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]
    gap = (s - last[q]) / 1e3 if q in last else float("nan")
    last[q] = e
    tot[q] += (e - s) / 1e3
    print(f"q{q} +{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:6.1f}  {short(r['Kernel_Name'])}")
print({q: round(v, 1) for q, v in tot.items()})
