#!/bin/bash
# GPU box: like tools/ab.sh for the 10^6-primitive scene (4 spp per step of 3840x2160).
# usage: tools/ab_synth.sh [-r rounds] name1 name2 ...   ("default" = the in-tree build)
ROUNDS=2
while getopts "r:" o; do case $o in r) ROUNDS=$OPTARG;; esac; done; shift $((OPTIND-1))
run() { timeout -k 10 170 python bench.py --steps 4 --spp 4 --warmup 2 --repeats 3 --no-cpu-baseline --no-configs --scene synth1m 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f Mpaths/s kernel %.3f ms'%(d['value'], d['roofline']['kernel_ms_avg']))"; }
for r in $(seq $ROUNDS); do for v in "$@"; do
  lib=""; [ "$v" != default ] && lib=$PWD/raytracing_amd/lib/variants/librt_amd_$v.so
  echo -n "synth1m $v : "; RT_AMD_LIB=$lib run || exit 1
done; done
