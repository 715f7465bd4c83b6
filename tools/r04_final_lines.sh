#!/bin/bash
# On the GPU box: the driver's default bench line (N times, $1, default 1) and the two model stats of the final tree.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/profiles; mkdir -p $OUT
for i in $(seq 1 ${1:-1}); do
  ( cd $R && timeout -s KILL 600 python bench.py > $OUT/r04_bench_line_$i.json 2> /tmp/bench_$i.err ) || tail -3 /tmp/bench_$i.err
  python - $OUT/r04_bench_line_$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
m = d.get("models", {})
print("value", d["value"], "x", d.get("speedup_vs_cpu_baseline"), "| 1%", d["escapes"]["0.01"]["value"],
      "| c1", m["c1"]["ms_per_step"], "c4", m["c4"]["ms_per_step"], "c1_f32", m.get("c1_f32", {}).get("ms_per_step"),
      "| traffic", d["roofline"]["traffic"])
PY
done
for wl in bls2017 bmshj2018; do
  rm -rf /tmp/st_$wl; timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d /tmp/st_$wl -- python $R/bench.py --workload $wl --steps 16 --warmup 2 --no-cpu-baseline > /tmp/st_$wl.log 2>&1
  python $R/tools/rocprof_summary.py /tmp/st_$wl $OUT/r04_${wl}_stats.md "Round 4: python bench.py --workload $wl --steps 16 --warmup 2 --no-cpu-baseline (rocprofv3 --kernel-trace --stats)" | head -3 || true
done
