#!/bin/bash
# On the GPU box: model steps over stream layouts (bench.py --partition / --model-depth / --coder-cus).
R=$(pwd)
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['config'].get('cu_partition'))"; }
for w in bmshj2018 bls2017; do
  for p in single plain masked; do
    for d in 2 3 4; do
      timeout 200 python bench.py --workload $w --partition $p --model-depth $d --steps 12 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | show "$w $p depth=$d"
    done
  done
  for p in pipelined pipelined-plain; do
    timeout 200 python bench.py --workload $w --partition $p --steps 12 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | show "$w $p"
  done
done
