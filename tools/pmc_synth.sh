#!/bin/bash
# GPU box: SQ counters of the render kernel on the 1 000 001-primitive scene (read from HBM).
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pmc_synth; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/a -- python3 $R/tools/bench_synth.py 3 > $O/a.log 2>&1 || { tail -n 5 $O/a.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -- python3 $R/tools/bench_synth.py 3 > $O/f.log 2>&1 || { tail -n 5 $O/f.log; exit 1; }
python3 $R/tools/pmc_summary.py $(find $O -name "*counter_collection.csv") | grep -A16 "render_mega_bvh<false, false"
