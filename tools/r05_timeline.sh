#!/bin/bash
# On the GPU box: start / end of every kernel of the last 20-batch group of the headline command (rocprofv3 kernel
# trace), relative to the group's first kernel: where a decode call's time goes beside its chain.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/tl; timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /tmp/tl.log 2>&1
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/tl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
rows.sort()
# the last launch of the decoder chain and everything from the encoder expansion in front of it
idx = max(i for i, r in enumerate(rows) if "dec_chain_kernel" in r[2])
start = max(i for i, r in enumerate(rows[:idx]) if "enc_expand_kernel" in r[2])
t0 = rows[start][0]
for s, e, n in rows[start - 6: idx + 12]:
    print("%9.3f %9.3f %8.3f ms  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n))
PY
