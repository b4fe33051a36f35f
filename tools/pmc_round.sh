#!/bin/bash
# GPU box: the profiles of the benchmarked kernels (tools/pmc_scene.sh each), kept apart by tag.
# usage: tools/pmc_round.sh <tag prefix> [all]     (all: the four small scenes of configs 2 and 3 as well)
P=${1:-r2}
R=${GRAFT_REPO_ROOT:-$PWD}
bash $R/tools/pmc_scene.sh ${P}_bouncing bouncing_spheres || exit 1
echo "---- cornell_box"
bash $R/tools/pmc_scene.sh ${P}_cornell cornell_box || exit 1
echo "---- synth1m"
bash $R/tools/pmc_scene.sh ${P}_synth1m synth1m --spp 8 --repeats 1 || exit 1
if [ "$2" = all ]; then
  for s in checkered_spheres perlin_spheres earth quads simple_light; do
    echo "---- $s"
    bash $R/tools/pmc_scene.sh ${P}_${s%%_spheres} $s || exit 1
  done
fi
