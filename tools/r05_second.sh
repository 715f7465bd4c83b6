#!/bin/bash
# Round 5, second call: the general SignalConv2D path on the kernels, the saturation curve with the larger temporaries
# budget, the cold copy reference.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_signal_conv_gpu.py -m gpu -x -q > gpurun_out/r05_conv_tests.log 2>&1; tail -25 gpurun_out/r05_conv_tests.log
timeout 600 python -m pytest tests -m gpu -x -q -k "range_coder or pipe or pipeline" > gpurun_out/r05_coder_tests.log 2>&1; tail -5 gpurun_out/r05_coder_tests.log
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r05_bench_b.log 2> gpurun_out/r05_bench_b.err; tail -c 300 gpurun_out/r05_bench_b.err
