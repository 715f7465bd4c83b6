#!/bin/bash
# on the GPU box: rebuild range_coder.hip with extra flags ($1) and run a python script ($2)
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $1 -c compression_amd/csrc/range_coder.hip -o build/range_coder.hip.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o compression_amd/libtfc_hip.so build/*.o || exit 1
python $2 2>&1 | grep -v amdgpu
