#!/bin/bash
# GPU box: the round's standard check: the GPU tests, then short benches of each workload.
# usage: tools/gpu_round.sh <tag> [pytest -k expression]
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
K=()
[ -n "$2" ] && K=(-k "$2")
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread "${K[@]}" > $O/tests.log 2>&1
rc=$?
tail -n 15 $O/tests.log
[ $rc -ne 0 ] && exit $rc
for sc in bouncing_spheres cornell_box perlin_spheres quads earth simple_light checkered_spheres; do
  timeout -k 10 200 python bench.py --scene $sc --steps 20 --warmup 3 --no-cpu-baseline --no-configs > $O/bench_$sc.json 2> $O/bench_$sc.err || { echo "bench $sc failed"; tail -n 5 $O/bench_$sc.err; exit 1; }
  python - $O/bench_$sc.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print(f'{d["config"]["scene"]:18s} {d["value"]:9.1f} Mpaths/s {d["mrays_per_s"]:9.1f} Mrays/s  kernel {r["kernel_ms_avg"]:.3f} ms alone, frac {r["frac"]:.4f}')
PY
done
