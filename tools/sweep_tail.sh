#!/bin/bash
# GPU box: sweep of the launch-tail knobs (paths a wave hands over, server workgroups) on the
# pipelined bench.  usage: tools/sweep_tail.sh [scene] ["spill_max values"] ["tail_blocks values"]
SCENE=${1:-bouncing_spheres}
SM=${2:-"0 8 16 24 32 48"}
TB=${3:-"8 16 32 64"}
STEPS=${STEPS:-100}; SPP=${SPP:-0}
run() { timeout -k 10 170 python bench.py --steps $STEPS --spp $SPP --warmup 2 --repeats 3 --no-cpu-baseline --no-configs --scene $SCENE 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f Mpaths/s kernel %.3f ms'%(d['value'], d['roofline']['kernel_ms_avg']))"; }
for sm in $SM; do for tb in $TB; do
  echo -n "spill_max $sm tail_blocks $tb : "; RT_AMD_SPILL_MAX=$sm RT_AMD_TAIL_BLOCKS=$tb run || exit 1
done; done
