#!/bin/bash
# GPU box: sweep of the decoupled scheduler's thresholds on the pipelined bench.
# usage: tools/sweep_sched.sh [scene] ["regen_min values"] ["min_ready values"] ["leaf_min values"]
SCENE=${1:-bouncing_spheres}
RG=${2:-"1 4 8 16"}
MR=${3:-"32 40 48"}
LM=${4:-"8"}
STEPS=${STEPS:-100}; SPP=${SPP:-0}
run() { timeout -k 10 170 python bench.py --steps $STEPS --spp $SPP --warmup 2 --repeats 3 --no-cpu-baseline --no-configs --scene $SCENE 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f Mpaths/s kernel %.3f ms'%(d['value'], d['roofline']['kernel_ms_avg']))"; }
for rg in $RG; do for mr in $MR; do for lm in $LM; do
  echo -n "regen_min $rg min_ready $mr leaf_min $lm : "; RT_AMD_REGEN_MIN=$rg RT_AMD_MIN_READY=$mr RT_AMD_LEAF_MIN=$lm run || exit 1
done; done; done
