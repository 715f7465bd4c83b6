#!/bin/bash
# On the GPU box: the C-ABI timing harness against each variant library under ab/ (same box, back to back, twice).
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for v in $R/ab/*/; do
  /opt/rocm/bin/hipcc -O2 -std=c++17 $R/tools/ubench/lanes_bench.cpp -I$R/include -L$v -ltfc_hip -Wl,-rpath,$v -o /tmp/lanes_bench_ab || exit 1
  echo "== $(basename $v)"
  NGROUPS=1 GPU_MAX_HW_QUEUES=16 timeout -s KILL 120 /tmp/lanes_bench_ab "$@" 2>&1 | grep -v amdgpu.ids
done
done
