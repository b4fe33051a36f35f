#!/bin/bash
# GPU box: rocprofv3 evidence for the bench workload.  usage: tools/profile_bench.sh <tag>
# Writes gpurun_out/<tag>/{kt,kt_serial,pmc1,pmc2,pmc_f,pmc_w}; each pass is a run of its own.
# kt: the default bench (consecutive launches overlap: the trace's kernel durations are spans of
# launches that share the GPU); kt_serial: the same with --serial (each launch alone: its
# average is the figure the bench line's roofline.kernel_ms_avg must agree with).
set -e
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 20 --warmup 3 --no-configs > $O/bench.json 2> $O/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_serial -- python3 $R/bench.py --steps 20 --warmup 3 --serial --no-cpu-baseline --no-configs > $O/bench_serial.json 2> $O/bench_serial.err
B="python3 $R/bench.py --steps 3 --warmup 1 --serial --no-cpu-baseline --no-configs"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmc1 -- $B > $O/pmc1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc2 -- $B > $O/pmc2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- $B > $O/pmc_f.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- $B > $O/pmc_w.log 2>&1
cat $O/bench.json $O/bench_serial.json
find $O/kt $O/kt_serial -name "*kernel_stats.csv" -exec head -5 {} \;
python3 $R/tools/pmc_summary.py $(find $O/pmc1 $O/pmc2 $O/pmc_f $O/pmc_w -name "*counter_collection.csv") > $O/pmc_summary.txt
cat $O/pmc_summary.txt
