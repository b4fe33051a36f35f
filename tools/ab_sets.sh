#!/bin/bash
# GPU box: pipelined launches rotating through two or three record sets (RT_AMD_SETS), alternated.
# usage: tools/ab_sets.sh [rounds]
R=${1:-3}
run() { timeout -k 10 200 python bench.py --steps ${STEPS:-200} --warmup 10 --repeats 3 --no-cpu-baseline --no-configs --scene $1 ${2:+--spp $2} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f Mpaths/s (spread %.2f %%), %.3f ms per step'%(d['value'], 100*d['repeat_spread'], d['ms_per_step']))"; }
for r in $(seq $R); do for sets in 2 3; do
  for sc in bouncing_spheres cornell_box perlin_spheres simple_light; do
    echo -n "sets $sets $sc : "; RT_AMD_SETS=$sets run $sc || exit 1
  done
  echo -n "sets $sets synth1m : "; STEPS=10 RT_AMD_SETS=$sets run synth1m 8 || exit 1
done; done
