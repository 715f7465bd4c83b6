#!/bin/bash
# On the GPU box: the round-3 profile set of the bench command (kernel stats, HBM traffic counters, SQ counters).
# Every rocprofv3 run is bounded; only summaries travel back (gpurun_out/profiles/).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/profiles; mkdir -p $OUT
# the driver's command (python bench.py --gpus 1 --steps 20 --warmup 5): 20 coding steps per lane-kernel launch
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
export TFC_PROFILE_STEPS_PER_LAUNCH=20
rm -rf /tmp/st; timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d /tmp/st -- python $R/bench.py $ARGS > /tmp/st.log 2>&1
tail -1 /tmp/st.log | cut -c1-300
python $R/tools/rocprof_summary.py /tmp/st $OUT/r03_bench_stats.md "Round 3: python bench.py --steps 20 --warmup 5 --no-cpu-baseline (rocprofv3 --kernel-trace --stats)" | head -14 || true
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  timeout -s KILL 150 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$ctr -- python $R/bench.py $ARGS > /tmp/pmc_$ctr.log 2>&1
  tail -1 /tmp/pmc_$ctr.log | cut -c1-200
done
python $R/tools/pmc_summary.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $OUT/r03_pmc_traffic | head -12
CTRS="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU"
rm -rf /tmp/sq
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d /tmp/sq -- python $R/bench.py $ARGS > /tmp/sq.log 2>&1
tail -1 /tmp/sq.log | cut -c1-200
python $R/tools/sq_summary.py /tmp/sq $OUT/r03_sq_inflight | head -8
# the model pipelines (BASELINE configs 1 and 4): kernel stats of a few steps
for wl in bls2017 bmshj2018; do
  rm -rf /tmp/st_$wl; timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d /tmp/st_$wl -- python $R/bench.py --workload $wl --steps 4 --warmup 2 --no-cpu-baseline > /tmp/st_$wl.log 2>&1
  tail -1 /tmp/st_$wl.log | cut -c1-200
  python $R/tools/rocprof_summary.py /tmp/st_$wl $OUT/r03_${wl}_stats.md "Round 3: python bench.py --workload $wl --steps 4 --warmup 2 --no-cpu-baseline (rocprofv3 --kernel-trace --stats)" | head -8 || true
done
# kernel-trace timelines of the model steps in flight (one character per bin and queue)
for cfg in "bmshj2018 1000 260 60 c4" "bls2017 250 70 15 c1"; do
  set -- $cfg
  rm -rf /tmp/tl_$1; timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$1 -- python $R/bench.py --workload $1 --steps 24 --warmup 2 --no-cpu-baseline > /tmp/tl_$1.log 2>&1
  f=$(find /tmp/tl_$1 -name "*kernel_trace.csv" | head -1)
  { grep "^{" /tmp/tl_$1.log | cut -c1-220; python $R/tools/trace_summary.py $f --bin $2 --last-ms $3 --skip-last-ms $4; } > $OUT/r03_$5_overlap_timeline.txt
  head -3 $OUT/r03_$5_overlap_timeline.txt | cut -c1-200
done
