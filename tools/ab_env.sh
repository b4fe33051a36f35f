#!/bin/bash
# GPU box: alternates short bench runs of one scene under different environment settings.
# usage: tools/ab_env.sh [-s scene] [-r rounds] [-a "extra bench arguments"] "NAME=VALUE ..." "NAME=VALUE ..." ...
# ("-" stands for no extra settings)
SCENE=synth1m; ROUNDS=2; ARGS="--steps 4 --spp 4 --warmup 2 --repeats 3"
while getopts "s:r:a:" o; do case $o in s) SCENE=$OPTARG;; r) ROUNDS=$OPTARG;; a) ARGS=$OPTARG;; esac; done; shift $((OPTIND-1))
for r in $(seq $ROUNDS); do for v in "$@"; do
  e=""; [ "$v" != "-" ] && e="$v"
  echo -n "$SCENE [$v] : "
  env $e timeout -k 10 170 python bench.py $ARGS --no-cpu-baseline --no-configs --scene $SCENE 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f Mpaths/s kernel %.3f ms'%(d['value'], d['roofline']['kernel_ms_avg']))" || exit 1
done; done
