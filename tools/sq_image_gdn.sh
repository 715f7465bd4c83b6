#!/bin/bash
# On the GPU box: SQ counters of the image-side layer's kernels (tools/image_gdn_layer_probe.py), two --pmc passes.
set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/profiles
export PYTHONPATH=$R
P=1
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
            "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  rm -rf /tmp/sq_img
  timeout -s KILL 240 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d /tmp/sq_img -- python $R/tools/image_gdn_layer_probe.py > /tmp/sq_img.log 2>&1
  tail -3 /tmp/sq_img.log | cut -c1-160
  python $R/tools/sq_summary.py /tmp/sq_img $R/gpurun_out/profiles/${1:-r04_sq_image_gdn}_$P 2>&1 | grep -i "conv_image\|kernel |" | cut -c1-600
  P=$((P+1))
done
