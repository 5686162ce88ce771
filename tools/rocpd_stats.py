#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 rocpd database (the default output when --output-format csv is not given):
   python tools/rocpd_stats.py <results.db> [steps] -> name, calls, total ms, mean us, share; with `steps`, per-step figures."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
kcols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
name_col = "display_name" if "display_name" in kcols else "kernel_name"
rows = list(cur.execute(f"select s.{name_col}, count(*), sum(d.end - d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"{'kernel':100s} {'calls':>8s} {'ms':>9s} {'mean us':>9s} {'share':>6s}")
for n, c, t in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    print(f"{n[:100]:100s} {c / steps:8.1f} {t / 1e6 / steps:9.3f} {t / c / 1e3:9.1f} {100 * t / tot:5.1f}%")
print(f"total {tot / 1e6 / steps:.2f} ms")
