#!/bin/bash
# GPU box: short benches of the linear scenes for a set of environment settings.
# usage: tools/ab_scenes.sh "VAR=1 VAR2=2" "..."   (each argument one setting; "A=0" = the defaults)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for cfg in "$@"; do
  echo "== $cfg"
  for sc in cornell_box quads simple_light earth checkered_spheres perlin_spheres; do
    env $cfg timeout -k 10 100 python bench.py --scene $sc --steps 20 --warmup 3 --no-cpu-baseline --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  %-18s %8.0f Mpaths/s kernel alone %.3f ms' % (d['config']['scene'], d['value'], d['roofline']['kernel_ms_avg']))" || exit 1
  done
done
