#!/bin/bash
# Here (no GPU): builds timing variants of the decoder chain whose every step carries two extra (dummy) instructions in one
# of its four parts — quotient (1), first shadow (2), second shadow (3), bounds (4) — into ab/; then on the GPU box
# `OVS=0 LINES=1 bash tools/r05_chain_ab.sh` gives the cycles per row of each: a part where two instructions
# cost ~8.6 cycles is bound by instruction issue, a part where they cost nothing sits under an LDS wait.
set -e -o pipefail
cd $(dirname $0)/..
rm -rf ab/pad*; mkdir -p ab
for part in ${PARTS:-0 1 2 3 4}; do
  bash tools/ab_lib.sh compression_amd/csrc ab/pad$part "-DTFC_PIPE_TIMING=1 -DTFC_PDEC_PAD=$part" > /tmp/ab_build.log 2>&1 || { tail -5 /tmp/ab_build.log; exit 1; }
  test -f ab/pad$part/libtfc_hip.so && echo built pad$part
done
