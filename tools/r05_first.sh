#!/bin/bash
# Round 5, first call: the whole GPU tier, then the driver's bench command (new sub-objects: saturation, cold GDN, c4_f32).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -s -k "gdn_f32" > gpurun_out/r05_gdn_f32.log 2>&1; tail -3 gpurun_out/r05_gdn_f32.log
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r05_pytest_a.log 2>&1; tail -5 gpurun_out/r05_pytest_a.log
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r05_bench_a.log 2> gpurun_out/r05_bench_a.err; tail -c 400 gpurun_out/r05_bench_a.err
tail -c 3000 gpurun_out/r05_bench_a.log
