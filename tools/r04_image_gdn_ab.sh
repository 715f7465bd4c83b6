#!/bin/bash
# The image-side layer with its GDN in one kernel (conv_image_gdn_kernel) against the two kernels, same box:
# the layer alone, then bmshj2018 (C4) bench lines (128 steps, as the default line's models.c4) with TFC_CONV_GDN_IMAGE = 1 / 0, alternating.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
  PYTHONPATH=. timeout 300 python tools/image_gdn_layer_probe.py 2>&1 | tail -4
  for rep in 1 2 3; do
    for sw in 1 0; do
      echo "== bmshj2018 TFC_CONV_GDN_IMAGE=$sw (run $rep)"
      TFC_CONV_GDN_IMAGE=$sw timeout 300 python bench.py --workload bmshj2018 --steps 128 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 |
        python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])"
    done
  done
} > gpurun_out/image_gdn_ab.txt 2>&1
cat gpurun_out/image_gdn_ab.txt
