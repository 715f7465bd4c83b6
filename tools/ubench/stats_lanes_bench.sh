#!/bin/bash
# On the GPU box: per-kernel durations of the harness (one group in flight).  Usage: stats_lanes_bench.sh <depth>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O2 -std=c++17 $R/tools/ubench/lanes_bench.cpp -I$R/include -L$R/compression_amd -ltfc_hip \
    -Wl,-rpath,$R/compression_amd -o /tmp/lanes_bench || exit 1
rm -rf /tmp/st_lb
NGROUPS=1 timeout -s KILL 90 rocprofv3 --kernel-trace --output-format csv -d /tmp/st_lb -- /tmp/lanes_bench 2 512 49152 $1 > /tmp/st_lb.log 2>&1
grep "steps per launch" /tmp/st_lb.log
python - <<PY
import csv, glob
from collections import defaultdict
d = defaultdict(list)
for f in glob.glob("/tmp/st_lb/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        d[row["Kernel_Name"][:60]].append(((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3, row.get("Grid_Size"), row.get("Workgroup_Size")))
for k, v in sorted(d.items(), key=lambda kv: -sum(x[0] for x in kv[1]))[:6]:
    big = sorted(v, key=lambda x: -x[0])[:3]
    print(k, len(v), "max3:", [(round(a), g, w) for a, g, w in big])
PY
