#!/bin/bash
# GPU box: the driver's own command (bench.py without flags) under rocprofv3 --kernel-trace --stats.
# Launches overlap in this run: the trace's render-kernel durations are spans of launches that share
# the GPU; the 8 isolated launches bench.py times for the roofline are among them.
# usage: tools/profile_default.sh <tag>
TAG=${1:-prof_default}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
$R/examples/bin/rt_scenes synth1m --half 500 --dump /tmp/bench_synth1m.rtscene > /dev/null || exit 1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py > $O/bench.json 2> $O/bench.err || { echo "failed"; tail -n 5 $O/bench.err; exit 1; }
find $O/kt -name "*kernel_stats.csv" -exec head -12 {} \; > $O/kernel_stats.csv
cut -c1-200 $O/kernel_stats.csv
tail -n 3 $O/bench.err
