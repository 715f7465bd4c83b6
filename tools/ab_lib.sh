#!/bin/bash
# Builds a variant libtfc_hip.so for same-box A/B timing: tools/ab_lib.sh <csrc dir> <out dir>
# (only range_coder.hip is recompiled from <csrc dir>; the other objects come from build/).
set -e
SRC=$1; OUT=$2
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $3 -c $SRC/range_coder.hip -o $OUT/range_coder.hip.o
OBJS=$(ls build/*.hip.o | grep -v range_coder.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $OUT/libtfc_hip.so $OUT/range_coder.hip.o $OBJS
rm $OUT/range_coder.hip.o
