#!/bin/bash
# GPU box: sweep of the tail hand-over parameters of pipelined launches.
run() { timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f Mpaths/s'%(d['value']))"; }
for cfg in "16 16" "32 16" "48 16" "63 16" "32 32" "48 32" "48 64" "8 16" "0 16"; do
  set -- $cfg
  echo -n "spill_max=$1 tail_blocks=$2 : "
  RT_AMD_SPILL_MAX=$1 RT_AMD_TAIL_BLOCKS=$2 run || exit 1
done
