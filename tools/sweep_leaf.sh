#!/bin/bash
# GPU box: sweep of the leaf-batch thresholds of the decoupled scheduler on the pipelined bench.
run() { timeout -k 10 170 python bench.py --steps 100 --warmup 2 --repeats 3 --no-cpu-baseline --no-configs --scene bouncing_spheres 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f Mpaths/s kernel %.3f ms'%(d['value'], d['roofline']['kernel_ms_avg']))"; }
for lm in 8 12 16 24; do for bm in 2 4 8; do echo -n "leaf_min $lm blocked_min $bm : "; RT_AMD_LEAF_MIN=$lm RT_AMD_BLOCKED_MIN=$bm run || exit 1; done; done
