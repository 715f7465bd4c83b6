#!/bin/bash
# On the GPU box: model steps over stream layouts (bench.py --partition / --model-depth / --coder-cus).
# Usage: bash tools/pipeline_sweep.sh "<workload> <partition> <depth> <coder_cus>" ...
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['config'].get('cu_partition'))"; }
for cfg in "$@"; do
  set -- $cfg
  timeout 200 python bench.py --workload $1 --partition $2 --model-depth $3 --coder-cus $4 --model-queue ${5:-2} --steps 32 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | show "$cfg"
done
