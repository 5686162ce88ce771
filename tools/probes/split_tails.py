"""Which split launches of a step wait for a finisher: parse the lab build's `[stamps wall]` / `[stamps BMxBN ...]` line pairs
(DLIP_LIB_PATH=.../libdeeplip_hip_lab.so DLIP_STAMP_PRINT=1 DLIP_STAMP_REDUCE_LATER=1 <eager step> 2> log) and list, per kernel
shape, launches, the kernel span and how much of it the median workgroup was NOT busy (= serial finisher + imbalance).
    python tools/probes/split_tails.py log"""
import re, sys, collections
rows = collections.OrderedDict()
wall = None
for ln in open(sys.argv[1], errors="replace"):
    m = re.search(r"\[stamps wall\] kernel span ([\d.]+) us; workgroup starts spread ([\d.]+) us; workgroup busy min ([\d.]+) med ([\d.]+) max ([\d.]+)", ln)
    if m:
        wall = tuple(float(v) for v in m.groups())
        continue
    m = re.search(r"\[stamps (\d+x\d+) M=(\d+) K=(\d+) nk=(\d+) G=(\d+)\]", ln)
    if m and wall:
        key = m.groups()
        r = rows.setdefault(key, [0, 0.0, 0.0, 0.0])
        r[0] += 1; r[1] += wall[0]; r[2] += wall[3]; r[3] += wall[0] - wall[3]
        wall = None
print(f"{'tile':8s} {'M':>7s} {'K':>5s} {'nk':>6s} {'G':>4s} {'n':>3s} {'span us':>9s} {'med busy':>9s} {'idle sum us':>11s}")
tot = 0.0
for (tile, M, K, nk, G), (n, span, med, idle) in sorted(rows.items(), key=lambda kv: -kv[1][3]):
    tot += idle
    print(f"{tile:8s} {M:>7s} {K:>5s} {nk:>6s} {G:>4s} {n:3d} {span / n:9.1f} {med / n:9.1f} {idle:11.1f}")
print(f"sum over all split launches of (span - median busy): {tot:.0f} us")
