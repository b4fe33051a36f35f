# bench the register-budget variants of the library (built with make WAVES=n)
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f Mpaths/s %.2f Grays/s kernel %.3f ms frac %.4f'%(d['value'], d['mrays_per_s']/1e3, d['roofline']['kernel_ms_avg'], d['roofline']['frac']))"; }
for scene in bouncing_spheres cornell_box perlin_spheres; do
for v in "" w6 w8; do
  for sched in 0 1; do
    [ "$scene" != bouncing_spheres ] && [ $sched = 1 ] && continue
    lib=""; [ -n "$v" ] && lib=$PWD/raytracing_amd/lib/variants/librt_amd_$v.so
    echo -n "$scene variant=${v:-w4} sched=$sched : "
    RT_AMD_LIB=$lib RT_AMD_SCHEDULER=$sched RT_AMD_MIN_READY=32 RT_AMD_LEAF_MIN=16 run --scene $scene --spp 64
  done
done; done
