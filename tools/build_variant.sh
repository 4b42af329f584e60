#!/bin/bash
# Builds a variant of libcude_hip.so with extra compiler flags into tools/abl_so/<name>.so (A/B runs on the GPU box:
# tools/abl_bench.py <name> ...).  usage: tools/build_variant.sh <name> [extra hipcc flags...]
# CUDE_SRC_ROOT=<checkout> builds another checkout's sources (e.g. a `git worktree` of the previous round's HEAD).
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=${CUDE_SRC_ROOT:-$ROOT}/conditional-ude_amd/csrc
OUT=$ROOT/tools/abl_so
TMP=$(mktemp -d)
mkdir -p $OUT
for f in cude_api cude_common cude_cpep cude_cpep2 cude_supp cude_adaptive; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable "$@" -c $SRC/$f.hip -o $TMP/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/$NAME.so $TMP/*.o -ldl
rm -rf $TMP
echo "built $OUT/$NAME.so"
