#!/bin/bash
# GPU box: sweep of the decoupled scheduler's thresholds on the bench workload
# (min_ready,leaf_min,blocked_min,regen_min per configuration; launches run alone).
run() { timeout -k 10 120 python bench.py --steps 20 --warmup 3 --serial --no-cpu-baseline --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f Mpaths/s kernel %.3f ms'%(d['value'], d['roofline']['kernel_ms_avg']))"; }
for cfg in ${@:-"40,8,2,1" "36,8,2,1" "44,8,2,1" "48,8,2,1" "40,12,2,1" "40,6,2,1" "40,8,4,1" "40,8,2,8" "40,8,2,16" "44,12,4,8" "40,8,2,1"}; do
  IFS=, read a b c d <<< "$cfg"
  echo -n "min_ready=$a leaf_min=$b blocked_min=$c regen_min=$d : "
  RT_AMD_MIN_READY=$a RT_AMD_LEAF_MIN=$b RT_AMD_BLOCKED_MIN=$c RT_AMD_REGEN_MIN=$d run || exit 1
done
