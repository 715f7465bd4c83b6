#!/bin/bash
# Round 5: the deferred-slab repair on the lane kernels (tools/r05_debug_outgrown.py), then the pipeline tests and the whole GPU tier.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
for cfg in "70 3000" "70 30"; do
  set -- $cfg
  echo "=== STREAMS=$1 SCALE=$2"
  STREAMS=$1 SCALE=$2 timeout 120 python tools/r05_debug_outgrown.py > gpurun_out/dbg_$1_$2.log 2>&1; echo rc=$?
  tail -8 gpurun_out/dbg_$1_$2.log
done
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05b_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r05b_pytest.log
