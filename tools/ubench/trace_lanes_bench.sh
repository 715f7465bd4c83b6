#!/bin/bash
# On the GPU box: kernel trace of the harness with D steps in flight; prints how many coder kernels overlap.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O2 -std=c++17 $R/tools/ubench/lanes_bench.cpp -I$R/include -L$R/compression_amd -ltfc_hip \
    -Wl,-rpath,$R/compression_amd -o /tmp/lanes_bench || exit 1
for D in "$@"; do
rm -rf /tmp/tr_lb
GPU_MAX_HW_QUEUES=${QUEUES:-16} timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_lb -- /tmp/lanes_bench 2 512 49152 $D > /tmp/tr_lb.log 2>&1
grep "in flight" /tmp/tr_lb.log
python - <<PY
import csv, glob
ev = []
for f in glob.glob("/tmp/tr_lb/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "lanes_kernel" in row["Kernel_Name"]:
            ev.append((int(row["Start_Timestamp"]), 1, row["Kernel_Name"][10:13], int(row.get("Queue_Id", 0) or 0)))
            ev.append((int(row["End_Timestamp"]), -1, "", 0))
ev.sort()
cur = peak = 0
hist = {}
last = ev[0][0]
for t, d, *_ in ev:
    hist[cur] = hist.get(cur, 0) + (t - last)
    last = t
    cur += d
    peak = max(peak, cur)
tot = sum(hist.values())
print("depth $D: peak concurrent coder kernels", peak, "time share by concurrency:",
      {k: round(v / tot, 3) for k, v in sorted(hist.items()) if v / tot > 0.01})
queues = sorted({e[3] for e in ev if e[1] == 1})
print("queues used:", len(queues))
PY
done
