R=${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 500 python -m pytest $R/tests/test_kernels_gpu.py -m gpu -x -q -k "rows or stats" 2>&1 | tail -2
for i in 1 2 3; do for L in A B; do
  if [ $L = A ]; then export DLIP_LIB_PATH=$R/deeplip_amd/lib/libdeeplip_hip_A.so; else unset DLIP_LIB_PATH; fi
  python3 $R/tools/probes/c3_chunked.py 2>&1 | grep "chunk 256" | head -1 | sed "s/^/lib $L: /"
  python3 $R/tools/bench_train_audio.py --batch 256 --steps 10 2>&1 | tail -1 | sed -E "s/.*: ([0-9.]+ ms\/step = [0-9.]+ utt\/s).*/lib $L audio train: \1/"
done; done
for L in A B; do
  if [ $L = A ]; then export DLIP_LIB_PATH=$R/deeplip_amd/lib/libdeeplip_hip_A.so; else unset DLIP_LIB_PATH; fi
  python3 $R/bench.py --no-cpu-baseline --single-mode --no-configs 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib $L headline', b['value'], b['roofline']['frac'], b['roofline']['kernels']['conv_rows_f16x3_kernel<160,256>'])"
done
