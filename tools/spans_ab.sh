#!/bin/bash
# Run ON THE GPU BOX: does letting the timed plan's MFMA launches time themselves in-kernel (dlip_span_scope_*, bench.py's default)
# cost the step anything?  Alternating runs, default vs --no-spans, three timed regions each (value_regions), same box.
# usage: tools/spans_ab.sh [rounds]      -> stdout (commit as profiles/rN/spans_ab.txt)
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-2}
echo "# bench.py --steps 20 --warmup 5 --no-configs --single-mode --no-cpu-baseline --no-h2d [--no-spans]; box $(python3 -c "import sys; sys.path.insert(0, \"$R\"); from deeplip_amd import build; print(build.box_id())"); $(date -u +%FT%TZ)"
for i in $(seq 1 $N); do
  for mode in spans no-spans; do
    extra=""; [ $mode = no-spans ] && extra="--no-spans"
    python3 $R/bench.py --steps 20 --warmup 5 --no-configs --single-mode --no-cpu-baseline --no-h2d $extra 2>/dev/null | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $i  %-9s value %9.1f  regions %s  ms/step %.4f  step_frac %.4f' % ('$mode', d['value'], d['value_regions'], d['ms_per_step'], d['roofline']['step_frac']))"
  done
done
