#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace of bench.py, keeps only the
# markdown summaries (the .db files are too large to travel back).
set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/profiles
run() {  # name, title, bench args...
  name=$1; title=$2; shift 2
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -- python $R/bench.py "$@" > /tmp/$name.log 2>&1
  grep '^{"metric"' /tmp/$name.log | tail -1 > $R/gpurun_out/profiles/$name.json
  python $R/tools/rocprof_summary.py /tmp/prof_$name $R/gpurun_out/profiles/$name.md "$title" > /dev/null
  echo "== $name"; head -c 1500 $R/gpurun_out/profiles/$name.json; echo; sed -n 5,12p $R/gpurun_out/profiles/$name.md
}
run "$@"
