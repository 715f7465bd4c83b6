#!/bin/bash
# On the GPU box: the round-6 profile set (kernel stats PER LEG of the default line, HBM traffic counters, SQ counters, the
# model pipelines, the driver's default line).  Every rocprofv3 run is bounded; only summaries travel back
# (gpurun_out/profiles/).  Counter passes serialise the kernels of a process, so they run with TFC_PIPE_OVERLAP=0: the chain
# behind the expansion, the parse behind the chain, every kernel with the chip to itself.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/profiles; mkdir -p $OUT
stats() {   # name, title, bench args...
  name=$1; title=$2; shift 2
  rm -rf /tmp/st_$name; timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d /tmp/st_$name -- python $R/bench.py "$@" > /tmp/st_$name.log 2>&1
  grep '^{' /tmp/st_$name.log | tail -1 | cut -c1-400
  python $R/tools/rocprof_summary.py /tmp/st_$name $OUT/$name.md "$title" | sed -n 5,12p || true
}
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
stats r06_headline_stats "Round 6, headline leg only: python bench.py $ARGS --leg headline (rocprofv3 --kernel-trace --stats): warm-up + 5 timed groups of 20 batches" $ARGS --leg headline
stats r06_single_batch_stats "Round 6, single-batch leg only: python bench.py $ARGS --leg single_batch (BASELINE config 2 as written: one 512-stream batch at a time, latency-mode handles)" $ARGS --leg single_batch
export TFC_PROFILE_STEPS_PER_LAUNCH=20
export TFC_PIPE_OVERLAP=0
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$ctr -- python $R/bench.py $ARGS --leg headline > /tmp/pmc_$ctr.log 2>&1
  tail -1 /tmp/pmc_$ctr.log | cut -c1-200
done
python $R/tools/pmc_summary.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $OUT/r06_pmc_traffic | head -14
# the same with the chain on the compact image (what launches of more than 24 batches run)
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcp_$ctr
  TFC_PIPE_FORMAT=pairs timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmcp_$ctr -- python $R/bench.py $ARGS --leg headline > /tmp/pmcp_$ctr.log 2>&1
done
python $R/tools/pmc_summary.py /tmp/pmcp_FETCH_SIZE /tmp/pmcp_WRITE_SIZE $OUT/r06_pmc_traffic_compact | head -8
CTRS="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU"
rm -rf /tmp/sq
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d /tmp/sq -- python $R/bench.py $ARGS --leg headline > /tmp/sq.log 2>&1
tail -1 /tmp/sq.log | cut -c1-200
python $R/tools/sq_summary.py /tmp/sq $OUT/r06_sq_inflight | head -10
unset TFC_PIPE_OVERLAP
unset TFC_PROFILE_STEPS_PER_LAUNCH
# GDN (BASELINE config 3): the counters of the default command's gdn leg
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcg_$ctr
  timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmcg_$ctr -- python $R/bench.py --leg gdn > /tmp/pmcg_$ctr.log 2>&1
done
python $R/tools/pmc_summary.py /tmp/pmcg_FETCH_SIZE /tmp/pmcg_WRITE_SIZE $OUT/r06_pmc_gdn | grep -i "gdn" | head -8
# the model pipelines (BASELINE configs 1 and 4): kernel stats of a few steps
for wl in bls2017 bmshj2018; do
  stats r06_${wl}_stats "Round 6: python bench.py --workload $wl --steps 16 --warmup 2 --no-cpu-baseline (rocprofv3 --kernel-trace --stats)" --workload $wl --steps 16 --warmup 2 --no-cpu-baseline
done
# the driver's default line of the same tree (bench_summary prints what the notes quote)
( cd $R && timeout -s KILL 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r06_bench_line.json 2> /tmp/bench.err ) || tail -5 /tmp/bench.err
python $R/tools/bench_summary.py $OUT/r06_bench_line.json 2>&1 | head -60
