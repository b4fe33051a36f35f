#!/bin/bash
# GPU box: wall time of the drop-in scene program from process start to the written PNG, next to the
# "Render took" line camera::render prints (flattening, scene creation, launches, PNG).
# usage: tools/e2e_times.sh
R=${GRAFT_REPO_ROOT:-$PWD}
export RTW_IMAGES=$R/tests/golden/images
cd /tmp
run() { local t0=$(date +%s%N); "$@" > /tmp/e2e.out 2>&1; local rc=$?; local t1=$(date +%s%N); echo "$(grep -o 'Render took [0-9.e+-]* seconds' /tmp/e2e.out) | wall $(( (t1 - t0) / 1000000 )) ms | rc $rc | ${*:2:3}"; }
for s in earth bouncing_spheres cornell_box; do for i in 1 2; do run $R/examples/bin/rt_scenes $s --spp 100 --out /tmp/o.png; done; done
run $R/examples/bin/rt_scenes synth1m --bvh 1 --spp 8 --out /tmp/o.png
RT_AMD_DEBUG=1 RTAMD_TIMING=1 $R/examples/bin/rt_scenes synth1m --bvh 1 --spp 8 --out /tmp/o.png 2>&1 | grep "rtamd\|librt_amd: scene"
RT_AMD_BUILD_DEVICE=1 RTAMD_TIMING=1 $R/examples/bin/rt_scenes synth1m --bvh 1 --spp 8 --out /tmp/o.png 2>&1 | grep rtamd
exit 0
