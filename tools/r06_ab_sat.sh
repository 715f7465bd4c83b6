#!/bin/bash
# On the GPU box: tools/r06_sat_probe.py under each variant library in ab/ (and the shipped one first).
# Usage: bash tools/r06_ab_sat.sh [points] [formats]
cd ${GRAFT_REPO_ROOT:-.}
cp compression_amd/libtfc_hip.so /tmp/libtfc_hip.keep
echo "== base"; python tools/r06_sat_probe.py ${1:-20,32} ${2:-0} 2>&1 | grep "^{" | cut -c1-330
for v in ab/*/; do
  [ -f $v/libtfc_hip.so ] || continue
  cp $v/libtfc_hip.so compression_amd/libtfc_hip.so
  echo "== $(basename $v)"; python tools/r06_sat_probe.py ${1:-20,32} ${2:-0} 2>&1 | grep "^{" | cut -c1-330
done
cp /tmp/libtfc_hip.keep compression_amd/libtfc_hip.so
