#!/bin/bash
# GPU box: throughput of bouncing_spheres against the samples per launch (how much the end of a launch costs).
set -e
mkdir -p gpurun_out/asym
for spp in 100 400 1600; do
  steps=$((20000/spp)); [ $steps -lt 6 ] && steps=6
  timeout -k 10 200 python bench.py --spp $spp --steps $steps --warmup 2 --repeats 3 --no-cpu-baseline --no-configs > gpurun_out/asym/b_$spp.json 2>gpurun_out/asym/b_$spp.err
  python - gpurun_out/asym/b_$spp.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"].get("spp_per_step"), d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"])
PY
done
