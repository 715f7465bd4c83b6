#!/bin/bash
# On the GPU box: the coder's parity tests (pipelined decoder included), then the chain clock probe (base library, then
# the variants under ab/).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_pipe_gpu.py tests/test_range_coder_gpu.py tests/test_pipeline_gpu.py tests/test_entropy_models_gpu.py -x -q 2>&1 | tail -6
OVS="0 2" bash tools/r05_chain_ab.sh
