R=$PWD
run() { python3 $R/bench.py --no-cpu-baseline --no-configs --single-mode "$@" 2>/dev/null | python3 -c "
import json, sys
b = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = b['roofline']; k = r['kernels']
print('$*'.ljust(22), b['value'], b['ms_per_step'], 'sum', r['kernels_ms_sum'], ' '.join(f\"{n.split('kernel')[-1]}:{v['tflops']:.0f}\" for n, v in k.items() if v['tflops'] > 50))"; }
for i in 1 2; do
  run
  run --dbg 5=0
  run --dbg 5=1
  run --dbg 3=0
  run --dbg 3=2
  run --dbg 4=1
done
