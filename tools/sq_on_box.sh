#!/bin/bash
# Runs on the GPU box (via gpurun): SQ issue / wait counters (8 slots = one pass) for the bench's
# kernels, serial and with steps in flight.  --pmc is combined with --kernel-trace only.
set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/profiles
CTRS="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU"
for mode in 1 6; do
  rm -rf /tmp/sq_$mode
  rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d /tmp/sq_$mode -- \
      python $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --inflight $mode > /tmp/sq_$mode.log 2>&1
  tail -1 /tmp/sq_$mode.log | cut -c1-200
  python $R/tools/sq_summary.py /tmp/sq_$mode $R/gpurun_out/profiles/${1:-r01_m}_sq_inflight$mode
done
