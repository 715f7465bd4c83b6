#!/bin/bash
# GDN / IGDN inside the convolution kernel by the size of the layer's output (TFC_CONV_GDN_MAX_MB), same box, steps in
# flight: bmshj2018 (C4, 128 steps) and bls2017 (C1, 32 steps), two runs of each setting, alternating.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
  for rep in 1 2; do
    for mb in 0 400 1536 100000; do
      echo "== bmshj2018 TFC_CONV_GDN_MAX_MB=$mb (run $rep)"
      TFC_CONV_GDN_MAX_MB=$mb timeout 300 python bench.py --workload bmshj2018 --steps 128 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 |
        python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])"
    done
    for mb in 0 1024; do
      echo "== bls2017 TFC_CONV_GDN_MAX_MB=$mb (run $rep)"
      TFC_CONV_GDN_MAX_MB=$mb timeout 300 python bench.py --workload bls2017 --steps 32 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 |
        python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])"
    done
  done
} > gpurun_out/gdn_fusion_ab.txt 2>&1
cat gpurun_out/gdn_fusion_ab.txt
