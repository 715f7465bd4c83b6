#!/bin/bash
# On the GPU box: SQ counters of the coder kernels under the C-ABI harness (one step alone).
# Usage: sq_lanes_bench.sh <dst-prefix under gpurun_out/> [mode]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O2 -std=c++17 $R/tools/ubench/lanes_bench.cpp -I$R/include -L$R/compression_amd -ltfc_hip \
    -Wl,-rpath,$R/compression_amd -o /tmp/lanes_bench || exit 1
mkdir -p $R/gpurun_out/profiles
CTRS="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU"
rm -rf /tmp/sq_lb
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d /tmp/sq_lb -- /tmp/lanes_bench ${2:-2} 512 49152 1 > /tmp/sq_lb.log 2>&1
tail -2 /tmp/sq_lb.log
python $R/tools/sq_summary.py /tmp/sq_lb $R/gpurun_out/profiles/$1
CTRS2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES SQ_INST_CYCLES_VMEM"
rm -rf /tmp/sq_lb2
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $CTRS2 --output-format csv -d /tmp/sq_lb2 -- /tmp/lanes_bench ${2:-2} 512 49152 1 > /tmp/sq_lb2.log 2>&1
python - <<PY
import sys
sys.path.insert(0, "$R/tools")
from pmc_summary import collect
c = collect("/tmp/sq_lb2")
for name, ctrs in c.items():
    if "lanes" in name or "fast" in name:
        print(name[:60], {k: round(sum(v) / len(v)) for k, v in ctrs.items()})
PY
