#!/bin/bash
# On the GPU box: time GDN backward on C3 ([262144, 192] bf16) with each variant library under ab/
# (built with tools/ab_lib.sh ... gdn_backward.hip), by copying it over compression_amd/libtfc_hip.so.
cd $GRAFT_REPO_ROOT
cp compression_amd/libtfc_hip.so /tmp/libtfc_hip.keep
for v in ab/*/; do
  cp $v/libtfc_hip.so compression_amd/libtfc_hip.so
  echo "== $(basename $v)"
  timeout -s KILL 300 python - <<'PY'
import torch, bench
r = bench.gdn_forward_bandwidth(torch.device("cuda:0"), steps=30)
print(r["backward"])
import subprocess, sys
sys.exit(subprocess.call([sys.executable, "-m", "pytest", "tests/test_gdn_gpu.py", "-m", "gpu", "-x", "-q"]))
PY
done
cp /tmp/libtfc_hip.keep compression_amd/libtfc_hip.so
