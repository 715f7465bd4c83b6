#!/bin/bash
# On the GPU box: HBM / fabric fetch and write sizes (rocprofv3 --pmc, separate passes, --kernel-trace only) of the
# second- and third-generation convolution kernels on the big layers of tools/conv3_check.py.
# Usage: bash tools/pmc_conv3.sh [case substring]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out/profiles
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_c3_$ctr
  timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_c3_$ctr -- python $R/tools/conv3_check.py 32 "${1:-192 @}" > /tmp/pmc_c3.log 2>&1
done
python $R/tools/pmc_summary.py /tmp/pmc_c3_FETCH_SIZE /tmp/pmc_c3_WRITE_SIZE $R/gpurun_out/profiles/${PMC_CONV3_NAME:-r06_pmc_conv3} | grep -i "conv\|kernel |" | cut -c1-200
