#!/bin/bash
# On the GPU box: rebuild signal_conv.hip with each tuning variant (CONV_CFGS, ';'-separated
# compiler flags) and run tools/conv_probe.py.
cd $GRAFT_REPO_ROOT
IFS=";" read -ra CFGS <<< "${CONV_CFGS:--DTFC_CONV_PF=4}"
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
for cfg in "${CFGS[@]}"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $cfg -c compression_amd/csrc/signal_conv.hip -o build/signal_conv.hip.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o compression_amd/libtfc_hip.so build/*.o || exit 1
  echo "== $cfg"
  python tools/conv_probe.py 2>&1 | grep -v amdgpu | head -7 | awk '{print $(NF-3), $(NF-2), $(NF-1)}' | paste -sd' '
done
