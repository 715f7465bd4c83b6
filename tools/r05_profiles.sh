#!/bin/bash
# On the GPU box: the round-5 profile set of the bench command (kernel stats, HBM traffic counters, SQ counters).
# Every rocprofv3 run is bounded; only summaries travel back (gpurun_out/profiles/).
# Counter passes serialise the kernels of a process, so they run with TFC_PIPE_OVERLAP=0: the chain behind the expansion,
# the parse behind the chain, every kernel with the chip to itself (what a per-kernel counter means anyway).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/profiles; mkdir -p $OUT
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
export TFC_PROFILE_STEPS_PER_LAUNCH=20
rm -rf /tmp/st; timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d /tmp/st -- python $R/bench.py $ARGS > /tmp/st.log 2>&1
tail -1 /tmp/st.log | cut -c1-300
python $R/tools/rocprof_summary.py /tmp/st $OUT/r05_bench_stats.md "Round 5: python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras (rocprofv3 --kernel-trace --stats)" | head -16 || true
export TFC_PIPE_OVERLAP=0
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  timeout -s KILL 150 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$ctr -- python $R/bench.py $ARGS > /tmp/pmc_$ctr.log 2>&1
  tail -1 /tmp/pmc_$ctr.log | cut -c1-200
done
python $R/tools/pmc_summary.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $OUT/r05_pmc_traffic | head -14
CTRS="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU"
rm -rf /tmp/sq
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d /tmp/sq -- python $R/bench.py $ARGS > /tmp/sq.log 2>&1
tail -1 /tmp/sq.log | cut -c1-200
python $R/tools/sq_summary.py /tmp/sq $OUT/r05_sq_inflight | head -10
unset TFC_PIPE_OVERLAP
# the model pipelines (BASELINE configs 1 and 4): kernel stats of a few steps
for wl in bls2017 bmshj2018; do
  rm -rf /tmp/st_$wl; timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d /tmp/st_$wl -- python $R/bench.py --workload $wl --steps 16 --warmup 2 --no-cpu-baseline > /tmp/st_$wl.log 2>&1
  tail -1 /tmp/st_$wl.log | cut -c1-200
  python $R/tools/rocprof_summary.py /tmp/st_$wl $OUT/r05_${wl}_stats.md "Round 5: python bench.py --workload $wl --steps 16 --warmup 2 --no-cpu-baseline (rocprofv3 --kernel-trace --stats)" | head -8 || true
done
# the driver's default line of the same tree (bench_summary prints what the notes quote)
( cd $R && timeout -s KILL 600 python bench.py > $OUT/r05_bench_line.json 2> /tmp/bench.err ) || tail -5 /tmp/bench.err
python $R/tools/bench_summary.py $OUT/r05_bench_line.json 2>&1 | head -60
