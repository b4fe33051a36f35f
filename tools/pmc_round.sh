#!/bin/bash
# GPU box: the profiles of the three benchmarked kernels (tools/pmc_scene.sh each), kept apart by tag.
# usage: tools/pmc_round.sh <tag prefix>
P=${1:-r2}
R=${GRAFT_REPO_ROOT:-$PWD}
bash $R/tools/pmc_scene.sh ${P}_bouncing bouncing_spheres || exit 1
echo "---- cornell_box"
bash $R/tools/pmc_scene.sh ${P}_cornell cornell_box || exit 1
echo "---- synth1m"
bash $R/tools/pmc_scene.sh ${P}_synth1m synth1m --spp 8 --repeats 1 || exit 1
