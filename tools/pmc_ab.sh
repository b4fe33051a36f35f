#!/bin/bash
# GPU box: SQ instruction counters of the render kernel for library variants (names under
# raytracing_amd/lib/variants, "default" = the in-tree build).  usage: tools/pmc_ab.sh [-s scene] name...
SCENE=bouncing_spheres
while getopts "s:" o; do case $o in s) SCENE=$OPTARG;; esac; done; shift $((OPTIND-1))
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=""; [ "$v" != default ] && lib=$R/raytracing_amd/lib/variants/librt_amd_$v.so
  O=$R/gpurun_out/pmc_ab/$v; rm -rf $O; mkdir -p $O
  RT_AMD_LIB=$lib timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O -- python3 $R/bench.py --steps 3 --warmup 1 --serial --no-cpu-baseline --no-configs --scene $SCENE > $O/log 2>&1 || { echo "$v failed"; tail -n 5 $O/log; exit 1; }
  echo "== $v"
  python3 $R/tools/pmc_summary.py $(find $O -name "*counter_collection.csv") | grep -A14 "render_mega_bvh<true, false\|render_mega<0, true, false>" | grep "INSTS\|ACTIVE\|GRBM\|vgpr"
done
