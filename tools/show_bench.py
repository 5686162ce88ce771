"""Print the headline fields of a bench.py JSON line (tools/show_bench.py FILE)."""
import json
import sys
d = json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
r = d["roofline"]
print("value", d["value"], "ms/step", d["ms_per_step"], "frac", r.get("frac"), "step_frac", r.get("step_frac"), "h2d", d.get("value_h2d_inclusive"))
for k, v in (r.get("kernels") or {}).items():
    print("  ", k, v)
for k in ("replayed_dominant", "replayed_spans"):
    if k in r:
        print(k, r[k])
for k, v in (d.get("configs") or {}).items():
    if isinstance(v, dict):
        print(k, {kk: vv for kk, vv in v.items() if not isinstance(vv, (dict, list))})
if "alt_mode" in d:
    print("alt_mode", {k: v for k, v in d["alt_mode"].items() if not isinstance(v, (dict, list))})
