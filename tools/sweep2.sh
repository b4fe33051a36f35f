run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f Mpaths/s %.2f Grays/s kernel %.3f ms'%(d['value'], d['mrays_per_s']/1e3, d['roofline']['kernel_ms_avg']))"; }
echo -n "sched=0 : "; RT_AMD_SCHEDULER=0 run
for cfg in "24 32 8" "32 32 8" "48 32 8" "32 16 4" "32 48 16" "40 40 12" "32 24 2" "32 64 64" "48 48 16" "56 32 8"; do
  set -- $cfg
  echo -n "sched=1 min_ready=$1 leaf_min=$2 blocked_min=$3 : "
  RT_AMD_SCHEDULER=1 RT_AMD_MIN_READY=$1 RT_AMD_LEAF_MIN=$2 RT_AMD_BLOCKED_MIN=$3 run
done
RT_AMD_SCHEDULER=1 RT_AMD_MIN_READY=32 python tools/sched_stats.py bouncing_spheres 100
