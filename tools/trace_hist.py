#!/usr/bin/env python
"""Busy milliseconds per queue in consecutive bins of a rocprofv3 --kernel-trace CSV (finding the timed region).
Usage: python tools/trace_hist.py kernel_trace.csv [--bin-ms 100]"""
import argparse
import collections
import csv

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--bin-ms", type=float, default=100.0)
a = ap.parse_args()
rows = []
with open(a.csv) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]))
t0 = min(r[0] for r in rows)
t1 = max(r[1] for r in rows)
bins = collections.defaultdict(lambda: collections.Counter())
for s, e, q in rows:
    bins[int((s - t0) / (a.bin_ms * 1e6))][q] += (e - s) * 1e-6
print(f"{len(rows)} kernels over {1e-6 * (t1 - t0):.1f} ms")
for b in sorted(bins):
    print(f"{b * a.bin_ms:9.0f} ms  " + "  ".join(f"q{q}:{v:6.1f}" for q, v in sorted(bins[b].items())))
