set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/profiles; mkdir -p $OUT
for wl in bls2017 bmshj2018; do
  rm -rf /tmp/st_$wl; timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d /tmp/st_$wl -- python $R/bench.py --workload $wl --steps 16 --warmup 2 --no-cpu-baseline > /tmp/st_$wl.log 2>&1
  tail -1 /tmp/st_$wl.log | cut -c1-200
  python $R/tools/rocprof_summary.py /tmp/st_$wl $OUT/r04_${wl}_stats.md "Round 4: python bench.py --workload $wl --steps 16 --warmup 2 --no-cpu-baseline (rocprofv3 --kernel-trace --stats)" | head -30 || true
done
