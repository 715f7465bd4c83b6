#!/bin/bash
# On the GPU box: SQ counters of the conv kernels at the C4 layer shapes (one --pmc pass, --kernel-trace only).
set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/profiles
CTRS="${CTRS:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES}"
rm -rf /tmp/sq_conv
timeout -s KILL 240 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d /tmp/sq_conv -- python $R/tools/conv_probe_c4.py ${BATCH:-32} > /tmp/sq_conv.log 2>&1
tail -12 /tmp/sq_conv.log | cut -c1-160
python $R/tools/sq_summary.py /tmp/sq_conv $R/gpurun_out/profiles/${1:-r02_sq_conv} 2>&1 | grep -i "conv\|kernel |" | cut -c1-400
