#!/bin/bash
# Builds raytracing_amd/lib/variants/librt_amd_<name>.so from the working tree or from a git ref.
# usage: tools/build_variant.sh <name> [git-ref|WORK] [extra make variables / CXXFLAGS via EXTRA=...]
set -e
NAME=$1; REF=${2:-WORK}; shift; shift || true
ROOT=$(cd $(dirname $0)/.. && pwd)
mkdir -p $ROOT/raytracing_amd/lib/variants
if [ "$REF" = WORK ]; then SRC=$ROOT; else
  SRC=/tmp/rt_variant_$NAME; rm -rf $SRC; git -C $ROOT worktree prune; git -C $ROOT worktree add -f --detach $SRC $REF >/dev/null
fi
B=/tmp/rt_build_$NAME; rm -rf $B; mkdir -p $B/raytracing_amd $B/include
cp -r $SRC/raytracing_amd/csrc $B/raytracing_amd/; cp $SRC/include/*.h $B/include/
make -C $B/raytracing_amd/csrc clean >/dev/null; make -C $B/raytracing_amd/csrc "$@" 2>&1 | grep -E "error|Error" || true
cp $B/raytracing_amd/lib/librt_amd.so $ROOT/raytracing_amd/lib/variants/librt_amd_$NAME.so
[ "$REF" = WORK ] || git -C $ROOT worktree remove --force $SRC
ls -la $ROOT/raytracing_amd/lib/variants/librt_amd_$NAME.so
