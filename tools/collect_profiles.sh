#!/bin/bash
# Authoring container: copies what tools/pmc_scene.sh left under gpurun_out/<tag> into profiles/ and
# derives the digest bench.py reads.  usage: tools/collect_profiles.sh <tag> [<tag> ...]
R=$(cd $(dirname $0)/.. && pwd)
for T in "$@"; do
  O=$R/gpurun_out/$T
  [ -f $O/pmc_summary.txt ] || { echo "no $O/pmc_summary.txt"; exit 1; }
  cp $O/kernel_stats.csv $R/profiles/${T}_serial_kernel_stats.csv
  tail -n 1 $O/bench_serial.json > $R/profiles/${T}_serial_bench.json
  cp $O/pmc_summary.txt $R/profiles/${T}_pmc.txt
  python3 $R/tools/make_traffic_json.py $O $R/profiles/${T}_hbm_traffic.json profiles/${T}_pmc.txt || exit 1
  echo "$T: $(head -2 $R/profiles/${T}_serial_kernel_stats.csv | tail -1 | cut -d, -f1-4 | cut -c1-120)"
done
