#!/bin/bash
# GPU box: SQ counters of one scene's render kernel for a library variant (RT_AMD_LIB), one pass.
# usage: tools/pmc_lib.sh <tag> <scene> <variant name or "default">
TAG=$1; SCENE=$2; V=$3
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
[ "$V" != default ] && export RT_AMD_LIB=$R/raytracing_amd/lib/variants/librt_amd_$V.so
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/a -- python3 $R/bench.py --steps 3 --warmup 1 --serial --no-cpu-baseline --no-configs --repeats 1 --scene $SCENE > $O/a.log 2>&1 || { tail -n 5 $O/a.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_IFETCH SQ_WAIT_IFETCH SQ_ACTIVE_INST_MISC --output-format csv -d $O/b -- python3 $R/bench.py --steps 3 --warmup 1 --serial --no-cpu-baseline --no-configs --repeats 1 --scene $SCENE > $O/b.log 2>&1 || { tail -n 5 $O/b.log; exit 1; }
python3 $R/tools/pmc_summary.py $(find $O -name "*counter_collection.csv") | grep -A24 "render_mega<0, true, false\|render_mega_bvh<true, false\|render_mega_bvh<false, false"
