set -e
timeout -k 10 200 python -u -m pytest tests/test_gpu_render.py tests/test_gpu_trace.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
for cfg in "0 24 1" "1 8 1" "1 16 1" "1 24 1" "1 32 1" "1 48 1" "1 24 8" "1 24 16" "1 32 16" "1 16 8"; do
  set -- $cfg
  echo -n "sched=$1 min_ready=$2 leaf_min=$3 : "
  RT_AMD_SCHEDULER=$1 RT_AMD_MIN_READY=$2 RT_AMD_LEAF_MIN=$3 timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f Mpaths/s %.2f Grays/s kernel %.3f ms frac %.4f'%(d['value'], d['mrays_per_s']/1e3, d['roofline']['kernel_ms_avg'], d['roofline']['frac']))"
done
