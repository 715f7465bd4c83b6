#!/bin/bash
# Runs on the GPU box (via gpurun): HBM traffic counters for the bench's kernels.
# FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950 (TCC has 4 slots, they cost 3 + 2),
# so each gets its own run; --pmc is combined with --kernel-trace only (MI355X_MICROARCH.md,
# "rocprofv3 PMC slots").  Only the per-kernel summary travels back.
set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/profiles
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$ctr -- \
      python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /tmp/pmc_$ctr.log 2>&1
  tail -2 /tmp/pmc_$ctr.log | cut -c1-300
done
python $R/tools/pmc_summary.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $R/gpurun_out/profiles/pmc
