#!/bin/bash
# Round 5, third call: device-side table build, deferred-slab retry, then the whole GPU tier.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_tables_gpu.py -m gpu -q -s > gpurun_out/r05_tables.log 2>&1; tail -40 gpurun_out/r05_tables.log
timeout 600 python -m pytest tests/test_pipeline_gpu.py -m gpu -q > gpurun_out/r05_pipeline.log 2>&1; tail -30 gpurun_out/r05_pipeline.log
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r05_pytest_c.log 2>&1; tail -15 gpurun_out/r05_pytest_c.log
