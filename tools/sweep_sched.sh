#!/bin/bash
# GPU box: sweep of the decoupled scheduler's thresholds on the bench workload.
run() { timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f Mpaths/s kernel %.3f ms'%(d['value'], d['roofline']['kernel_ms_avg']))"; }
for cfg in "40 8 2" "40 4 1" "40 1 1" "44 8 2" "44 4 1" "48 8 2" "44 6 2" "40 8 2"; do
  set -- $cfg
  echo -n "min_ready=$1 leaf_min=$2 blocked_min=$3 : "
  RT_AMD_MIN_READY=$1 RT_AMD_LEAF_MIN=$2 RT_AMD_BLOCKED_MIN=$3 run || exit 1
done
