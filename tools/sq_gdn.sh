#!/bin/bash
# On the GPU box: SQ issue / wait counters of the GDN kernels on C3 (one --pmc pass, --kernel-trace only).
# Usage: tools/sq_gdn.sh <dst prefix under gpurun_out/profiles>
set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/profiles
CTRS="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16"
rm -rf /tmp/sq_gdn
cat > /tmp/gdn_run.py <<PY
import sys, torch
sys.path.insert(0, "$R")
import bench
print(bench.gdn_forward_bandwidth(torch.device("cuda:0"), steps=5))
PY
timeout -s KILL 240 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d /tmp/sq_gdn -- python /tmp/gdn_run.py > /tmp/sq_gdn.log 2>&1
tail -2 /tmp/sq_gdn.log | cut -c1-400
python $R/tools/sq_summary.py /tmp/sq_gdn $R/gpurun_out/profiles/${1:-r02_sq_gdn} 2>&1 | grep -i "gdn\|kernel |"
