#!/bin/bash
# On the GPU box: the chain clock probe (first chain workgroup of the 20-batch launch) under each variant library in ab/.
cd ${GRAFT_REPO_ROOT:-.}
cp compression_amd/libtfc_hip.so /tmp/libtfc_hip.keep
run() { for ov in ${OVS:-0}; do TFC_PIPE_OVERLAP=$ov timeout 120 python tools/chain_clock_probe.py 2>&1 | grep "^overlap\|dec chain (build\|enc chain (TFC" | tail -${LINES:-1}; done; }
if [ -z "${SKIP_BASE:-}" ]; then echo "== base"; run; fi
for v in ab/*/; do
  [ -f $v/libtfc_hip.so ] || continue
  cp $v/libtfc_hip.so compression_amd/libtfc_hip.so
  echo "== $(basename $v)"; LINES=4 run
done
cp /tmp/libtfc_hip.keep compression_amd/libtfc_hip.so
