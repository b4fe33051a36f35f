#!/bin/bash
# GPU box: alternates bench runs of library variants (names under raytracing_amd/lib/variants,
# "default" = the in-tree build).  usage: tools/ab.sh [-s scene] [-r rounds] name1 name2 ...
SCENE=bouncing_spheres; ROUNDS=3
while getopts "s:r:" o; do case $o in s) SCENE=$OPTARG;; r) ROUNDS=$OPTARG;; esac; done; shift $((OPTIND-1))
run() { timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-configs --scene $SCENE 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f Mpaths/s kernel %.3f ms'%(d['value'], d['roofline']['kernel_ms_avg']))"; }
for r in $(seq $ROUNDS); do for v in "$@"; do
  lib=""; [ "$v" != default ] && lib=$PWD/raytracing_amd/lib/variants/librt_amd_$v.so
  echo -n "$SCENE $v : "; RT_AMD_LIB=$lib run || exit 1
done; done
