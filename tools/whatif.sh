#!/bin/bash
# Authoring container: builds the what-if variants of the library (extra work in the node step that
# changes no result), for tools/ab.sh on the GPU box: which resource the node step is short of.
set -e
cd $(dirname $0)/..
tools/build_variant.sh lds3 WORK -j4 EXTRA=-DRT_WHATIF_LDS_READS=3
tools/build_variant.sh minmax2 WORK -j4 EXTRA=-DRT_WHATIF_MINMAX=2
tools/build_variant.sh fma3 WORK -j4 EXTRA=-DRT_WHATIF_FMA=3
