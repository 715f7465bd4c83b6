#!/bin/bash
# On the GPU box: the pipelined coder's parity tests, then the headline command under the three ways of launching the
# encoder's chain (TFC_PIPE_OVERLAP), on the 8(d) law and on its 1 % variant.  Lines go to gpurun_out/r04_coder_check.log.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT
LOG=$OUT/r04_coder_check.log; : > $LOG
cd $R
timeout 420 python -m pytest tests/test_pipe_gpu.py tests/test_range_coder_gpu.py tests/test_pipeline_gpu.py tests/test_entropy_models_gpu.py -x -q 2>&1 | tail -15 | tee -a $LOG
line() {  # label, env assignment, bench args
  label=$1; shift; envs=$1; shift
  env $envs timeout 200 python bench.py --no-extras --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$label', 'value', d['value'], 'ms', d['ms_per_step'], 'inflight', json.dumps(d.get('kernels_ms_in_flight')))" 2>&1 | tee -a $LOG
}
for ov in 2 1 0; do line "overlap=$ov law8d" TFC_PIPE_OVERLAP=$ov; done
for ov in 2 0; do line "overlap=$ov 1pct" TFC_PIPE_OVERLAP=$ov --escape-fraction 0.01; done
line "overlap=2 law8d again" TFC_PIPE_OVERLAP=2
