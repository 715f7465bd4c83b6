#!/bin/bash
# On the GPU box: SQ counters (two --pmc passes, --kernel-trace only) of the convolution kernels on the big layers of
# tools/conv3_check.py.  Usage: bash tools/sq_conv3.sh <dst-prefix> [case substring]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out/profiles
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS"
P2="SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA"
i=1
for CTRS in "$P1" "$P2"; do
  rm -rf /tmp/sq_c3
  timeout -s KILL 240 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d /tmp/sq_c3 -- python $R/tools/conv3_check.py 32 "${2:-x2 192 @192}" > /tmp/sq_c3.log 2>&1
  tail -3 /tmp/sq_c3.log | cut -c1-200
  python $R/tools/sq_summary.py /tmp/sq_c3 $R/gpurun_out/profiles/${1:-sq_conv3}_p$i 2>&1 | grep -i "conv3\|conv_bf16\|kernel |" | cut -c1-500
  i=$((i+1))
done
