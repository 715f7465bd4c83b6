#!/bin/bash
# On the GPU box: tools/conv_probe_c4.py with each variant library under ab/ (tools/ab_lib.sh ... signal_conv.hip).
cd $GRAFT_REPO_ROOT
cp compression_amd/libtfc_hip.so /tmp/libtfc_hip.keep
for v in ab/*/; do
  cp $v/libtfc_hip.so compression_amd/libtfc_hip.so
  echo "== $(basename $v)"
  timeout -s KILL 200 python tools/conv_probe_c4.py ${BATCH:-128} 2>&1 | grep -v amdgpu.ids | cut -c1-45,100-160
done
cp /tmp/libtfc_hip.keep compression_amd/libtfc_hip.so
