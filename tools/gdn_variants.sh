#!/bin/bash
# On the GPU box: rebuild gdn.hip with each forward-kernel variant and time GDN forward on C3.
cd $GRAFT_REPO_ROOT
IFS=";" read -ra CFGS <<< "${GDN_CFGS:--DTFC_GDN_XPREFETCH=1}"
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
for cfg in "${CFGS[@]}"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $cfg -c compression_amd/csrc/gdn.hip -o build/gdn.hip.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o compression_amd/libtfc_hip.so build/*.o || exit 1
  echo "== $cfg"
  python - <<'PY'
import torch, bench
from compression_amd import _lib
for _ in range(2):
    r = bench.gdn_forward_bandwidth(torch.device("cuda:0"), steps=30)
print(r["kernel_ms"], r["achieved"])
PY
done
