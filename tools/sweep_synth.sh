#!/bin/bash
# GPU box: the 1 000 001-primitive scene under different workgroup / LDS-stack geometries.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/sweep_synth; mkdir -p $O
$R/examples/bin/rt_scenes synth1m --half 500 --dump /tmp/bench_synth1m.rtscene > /dev/null || exit 1
run() { echo -n "$* : "; env "$@" RT_AMD_DEBUG=1 timeout -k 10 150 python bench.py --scene synth1m --spp 8 --steps 3 --warmup 1 --repeats 3 --no-cpu-baseline --no-configs 2> $O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f Mpaths/s kernel %.3f ms'%(d['value'], d['roofline']['kernel_ms_avg']))" || { tail -n 5 $O/err.log; exit 1; }; grep -m1 "render geometry" $O/err.log; }
run A=0 || exit 1
for cfg in "$@"; do run $cfg || exit 1; done
