#!/bin/bash
# GPU box: tests, soak runs (every image of a scene equal to the first), A/B of library variants and
# the wave-level schedule statistics, each step under its own time limit; stops at the first failure.
# usage: tools/gpu_session.sh <tag> <variants for the A/B ("default" = in-tree)...>
TAG=${1:-session}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
step() { echo "== $*" | tee -a $O/session.log; }
if [ -z "$SKIP_TESTS" ]; then
step tests
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $O/tests.log 2>&1 || { tail -n 30 $O/tests.log; exit 1; }
tail -n 3 $O/tests.log
fi
if [ -z "$SKIP_SOAK" ]; then
step soak synth1m
timeout -k 10 400 python -u tools/soak.py synth1m 4 ${SOAK_SYNTH:-800} > $O/soak_synth1m.log 2>&1 || { tail -n 5 $O/soak_synth1m.log; exit 1; }
tail -n 2 $O/soak_synth1m.log
step soak bouncing_spheres
timeout -k 10 300 python -u tools/soak.py bouncing_spheres 100 ${SOAK_BOUNCING:-4000} > $O/soak_bouncing.log 2>&1 || { tail -n 5 $O/soak_bouncing.log; exit 1; }
tail -n 2 $O/soak_bouncing.log
fi
if [ $# -gt 0 ]; then
  step ab bouncing
  timeout -k 10 600 tools/ab.sh -r 3 "$@" 2>&1 | tee $O/ab_bouncing.log || exit 1
  step ab synth1m
  timeout -k 10 900 tools/ab_synth.sh -r 2 "$@" 2>&1 | tee $O/ab_synth.log || exit 1
fi
step sched stats
timeout -k 10 200 python tools/sched_stats.py bouncing_spheres 100 > $O/sched_bouncing.log 2>&1 || { tail -n 5 $O/sched_bouncing.log; exit 1; }
cat $O/sched_bouncing.log
timeout -k 10 300 python tools/sched_stats.py synth1m 4 > $O/sched_synth1m.log 2>&1 || { tail -n 5 $O/sched_synth1m.log; exit 1; }
cat $O/sched_synth1m.log
