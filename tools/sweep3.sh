run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f Mpaths/s %.2f Grays/s kernel %.3f ms'%(d['value'], d['mrays_per_s']/1e3, d['roofline']['kernel_ms_avg']))"; }
echo -n "sched=0 : "; RT_AMD_SCHEDULER=0 run
for u in 2 4 8; do
for cfg in "32 32 8" "32 16 4" "48 32 8" "24 24 4" "40 24 8"; do
  set -- $cfg
  echo -n "unroll=$u min_ready=$1 leaf_min=$2 blocked_min=$3 : "
  RT_AMD_LIB=$PWD/raytracing_amd/lib/variants/librt_amd_u$u.so RT_AMD_SCHEDULER=1 RT_AMD_MIN_READY=$1 RT_AMD_LEAF_MIN=$2 RT_AMD_BLOCKED_MIN=$3 run
done; done
