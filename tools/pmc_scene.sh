#!/bin/bash
# GPU box: kernel trace + the PMC passes (SQ instruction mix, SQ waits / LDS, L1 / L2 requests, HBM fetch, HBM write)
# of the render kernel for one workload, each pass a run of its own.
# usage: tools/pmc_scene.sh <tag> <scene> [extra bench.py arguments]
TAG=$1; SCENE=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
[ "$SCENE" = synth1m ] && { $R/examples/bin/rt_scenes synth1m --half 500 --dump /tmp/bench_synth1m.rtscene > /dev/null || exit 1; }
B="python3 $R/bench.py --steps 3 --warmup 1 --serial --no-cpu-baseline --no-configs --scene $SCENE $*"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 10 --warmup 2 --serial --no-cpu-baseline --no-configs --scene $SCENE "$@" > $O/bench_serial.json 2> $O/bench_serial.err || { echo "kernel trace failed"; tail -n 5 $O/bench_serial.err; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmc1 -- $B > $O/pmc1.log 2>&1 || { echo "pmc1 failed"; tail -n 5 $O/pmc1.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc2 -- $B > $O/pmc2.log 2>&1 || { echo "pmc2 failed"; tail -n 5 $O/pmc2.log; exit 1; }
# vector L1 and L2: requests the L1 sends on to L2, L1 accesses, L2 hits / misses, L2 reads from memory
timeout -s KILL 120 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/pmc3 -- $B > $O/pmc3.log 2>&1 || { echo "pmc3 failed"; tail -n 5 $O/pmc3.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- $B > $O/pmc_f.log 2>&1 || { echo "pmc_f failed"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- $B > $O/pmc_w.log 2>&1 || { echo "pmc_w failed"; exit 1; }
find $O/kt -name "*kernel_stats.csv" -exec head -6 {} \; > $O/kernel_stats.csv
cat $O/kernel_stats.csv
python3 $R/tools/pmc_summary.py $(find $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc_f $O/pmc_w -name "*counter_collection.csv") > $O/pmc_summary.txt
grep -v "reduce_stage" $O/pmc_summary.txt | head -60
