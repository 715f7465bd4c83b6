#!/bin/bash
# On the GPU box: time GDN forward on C3 ([262144, 192] bf16) with each variant library under ab/ (built with
# tools/ab_lib.sh ... gdn.hip), by copying it over compression_amd/libtfc_hip.so; twice round-robin.
cd $GRAFT_REPO_ROOT
cp compression_amd/libtfc_hip.so /tmp/libtfc_hip.keep
for round in 1 2; do
for v in ab/*/; do
  cp $v/libtfc_hip.so compression_amd/libtfc_hip.so
  echo "== $(basename $v)"
  timeout -s KILL 120 python - <<'PY'
import torch, bench
for _ in range(2):
    r = bench.gdn_forward_bandwidth(torch.device("cuda:0"), steps=30)
print({k: r[k] for k in ("kernel_ms", "achieved", "frac")})
PY
done
done
cp /tmp/libtfc_hip.keep compression_amd/libtfc_hip.so
