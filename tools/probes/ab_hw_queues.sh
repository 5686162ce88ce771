for i in 1 2 3; do for q in "GPU_MAX_HW_QUEUES=1" "GPU_MAX_HW_QUEUES=2" "GPU_MAX_HW_QUEUES=3" ""; do env $q python3 bench.py --no-cpu-baseline --single-mode --no-configs 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench [$q]', b['value'], b['ms_per_step'], b['roofline']['frac'], b['roofline'].get('step_frac'))"; done; done
