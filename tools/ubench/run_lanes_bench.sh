#!/bin/bash
# On the GPU box: build and run the C-ABI timing harness.  Usage: run_lanes_bench.sh [mode] [streams] [elems] [depths]
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
/opt/rocm/bin/hipcc -O2 -std=c++17 $R/tools/ubench/lanes_bench.cpp -I$R/include -L$R/compression_amd -ltfc_hip \
    -Wl,-rpath,$R/compression_amd -o /tmp/lanes_bench
GPU_MAX_HW_QUEUES=16 timeout -s KILL 120 /tmp/lanes_bench "$@"
