#!/usr/bin/env python
"""Every kernel between the end of a long decode kernel and the start of the next long encode kernel in a rocprofv3
--kernel-trace CSV (what the coder stream does between two steps of a model pipeline).
Usage: python tools/trace_gap.py kernel_trace.csv [--which -2]"""
import argparse
import csv

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--which", type=int, default=-2)
ap.add_argument("--min-ms", type=float, default=20.0)
a = ap.parse_args()
rows = []
with open(a.csv) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
rows.sort()
long_dec = [r for r in rows if "dec_" in r[3] and r[1] - r[0] > a.min_ms * 1e6]
d = long_dec[a.which]
nxt = [r for r in rows if "enc_" in r[3] and r[1] - r[0] > 0.4 * a.min_ms * 1e6 and r[0] > d[1]]
lo, hi = d[1] - 1_000_000, (nxt[0][0] if nxt else d[1] + 20_000_000) + 500_000
print(f"decode kernel ends at 0; next long encode starts at {1e-6 * ((nxt[0][0] if nxt else 0) - d[1]):.3f} ms")
for s, e, q, n in rows:
    if e >= lo and s <= hi:
        print(f"{1e-6 * (s - d[1]):9.3f} {1e-6 * (e - d[1]):9.3f}  q{q}  {n[:70]}")
