#!/bin/bash
# Prints VGPR / SGPR / spill counts of every kernel in a .hip file (gfx950 device code only).
# usage: tools/kernel_resources.sh conditional-ude_amd/csrc/cude_cpep.hip
set -e
SRC=$1
DIR=$(dirname $SRC)
TMP=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -I$DIR $SRC -o $TMP/k.s 2>/dev/null
grep -E "^\s+\.(vgpr_count|sgpr_count|sgpr_spill_count|vgpr_spill_count|name):" $TMP/k.s | paste - - - - - | sed 's/ \+/ /g'
rm -rf $TMP
