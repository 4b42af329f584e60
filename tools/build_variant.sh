#!/bin/bash
# Builds a variant of libcude_hip.so with extra compiler flags into tools/abl_so/<name>.so (A/B runs on the GPU box:
# tools/abl_bench.py <name> ...).  usage: [ONLY="cude_supp ..."] tools/build_variant.sh <name> [extra hipcc flags...]
# ONLY: recompile just these sources with the flags and take the other objects from the default build in csrc/ (for a
# flag that touches one file).  CUDE_SRC_ROOT=<checkout> builds another checkout's sources (e.g. a `git worktree` of the
# previous round's HEAD).
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=${CUDE_SRC_ROOT:-$ROOT}/conditional-ude_amd/csrc
OUT=$ROOT/tools/abl_so
TMP=$(mktemp -d)
mkdir -p $OUT
ALL=$(sed -n 's/^SRCS *= *//p' $SRC/Makefile | sed 's/\.hip//g')
if [ -n "$ONLY" ]; then
  make -s -C $SRC -j8
  for f in $ALL; do cp $SRC/$f.o $TMP/$f.o; done
  cp $SRC/*_p[0-9].o $TMP/          # (shape groups compiled as translation units of their own: Makefile PARTS)
  LIST=$ONLY
else
  LIST=$ALL
fi
for f in $LIST; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable "$@" -c $SRC/$f.hip -o $TMP/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/$NAME.so $TMP/*.o -ldl
rm -rf $TMP
echo "built $OUT/$NAME.so"
