#!/bin/bash
# Builds a variant libtfc_hip.so for same-box A/B timing:
#   tools/ab_lib.sh <csrc dir> <out dir> ["extra hipcc flags"] [translation unit, default range_coder.hip]
# (only that translation unit is recompiled from <csrc dir>; the other objects come from build/).
set -e
SRC=$1; OUT=$2; TU=${4:-range_coder.hip}
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $3 -c $SRC/$TU -o $OUT/$TU.o
OBJS=$(ls build/*.hip.o | grep -v "/$TU.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $OUT/libtfc_hip.so $OUT/$TU.o $OBJS
rm $OUT/$TU.o
