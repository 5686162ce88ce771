"""Top kernels of a rocprofv3 --kernel-trace --stats CSV directory: python tools/probes/stats_top.py DIR [passes] [n]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
passes = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 18]:
    print("%-100s calls/pass %6.1f  avg %8.1f us  %5.1f%%" % (r["Name"][:100], int(r["Calls"]) / passes, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print("total per pass: %.3f ms" % (tot / passes / 1e6))
