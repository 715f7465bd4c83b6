#!/bin/bash
# On the GPU box, first thing of round 5: the full GPU test tier, the driver's default bench line, and the
# kernel-stats profile of the headline command.  Summaries under gpurun_out/profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/profiles; mkdir -p $OUT
TAG=${1:-r05a}
( cd $R && timeout -s KILL 900 python -m pytest tests -m gpu -x -q > $R/gpurun_out/${TAG}_pytest.log 2>&1 ); echo "pytest rc=$?"; tail -4 $R/gpurun_out/${TAG}_pytest.log
( cd $R && timeout -s KILL 600 python bench.py > $OUT/${TAG}_bench_line.json 2> /tmp/bench.err ) || tail -5 /tmp/bench.err
python $R/tools/bench_summary.py $OUT/${TAG}_bench_line.json 2>&1 | head -40
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
rm -rf /tmp/st; timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d /tmp/st -- python $R/bench.py $ARGS > /tmp/st.log 2>&1
python $R/tools/rocprof_summary.py /tmp/st $OUT/${TAG}_bench_stats.md "Round 5 ($TAG): python bench.py $ARGS (rocprofv3 --kernel-trace --stats)" | head -16 || true
