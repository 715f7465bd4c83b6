#!/bin/bash
# On the GPU box: the C4 bench line with each variant library under ab/.
cd $GRAFT_REPO_ROOT
cp compression_amd/libtfc_hip.so /tmp/libtfc_hip.keep
for v in ab/*/; do
  cp $v/libtfc_hip.so compression_amd/libtfc_hip.so
  echo "== $(basename $v)"
  timeout -s KILL 200 python bench.py --workload bmshj2018 --steps ${STEPS:-4} --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels_ms_per_step'])"
done
cp /tmp/libtfc_hip.keep compression_amd/libtfc_hip.so
