#!/bin/bash
# HERE (no GPU): builds libtfc_hip.so variants with TFC_CONV3_EXP=<mask> into tools/probe_libs/ (git-ignored, travels
# with gpurun).  ON THE GPU BOX: `bash tools/conv3_variants.sh run [batch] [case]` times tools/conv3_check.py with each.
# Usage: bash tools/conv3_variants.sh build 1 2 4 ...
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
D=$R/tools/probe_libs
if [ "$1" = build ]; then
  shift; mkdir -p $D
  OBJS=$(ls $R/build/*.o | grep -v signal_conv.hip.o)
  for m in "$@"; do
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DTFC_CONV3_EXP=$m -c $R/compression_amd/csrc/signal_conv.hip -o /tmp/sc_exp$m.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $D/libtfc_conv3_exp$m.so $OBJS /tmp/sc_exp$m.o && echo built $m ) &
  done
  wait
else
  shift
  for lib in $D/libtfc_conv3_exp*.so; do
    echo "== $(basename $lib)"
    TFC_LIB_PATH=$lib timeout 120 python $R/tools/conv3_check.py ${1:-32} "${2:-192 @}" 2>&1 | grep -v amdgpu | cut -c1-100
  done
fi
