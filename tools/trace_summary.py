#!/usr/bin/env python
"""Timeline summary of a rocprofv3 --kernel-trace CSV: per queue, busy time, first start / last end, and a
coarse Gantt (one character per `--bin` microseconds, letter = first letter class of the kernel running).
Shows whether the coder stream and the transform stream of a model pipeline overlap.
Usage: python tools/trace_summary.py kernel_trace.csv [--bin 500] [--from-ms A --to-ms B]"""
import argparse
import csv
import collections


def klass(name):
    n = name.lower()
    if "conv" in n:
        return "C"
    if "gdn" in n:
        return "G"
    if "dec_" in n:
        return "D"
    if "enc_" in n:
        return "E"
    return "."


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--bin", type=float, default=500.0)
    ap.add_argument("--last-ms", type=float, default=400.0, help="only the last N milliseconds of the trace")
    ap.add_argument("--skip-last-ms", type=float, default=0.0, help="... ending this long before the last kernel")
    args = ap.parse_args()
    rows = []
    with open(args.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
    rows.sort()
    t_end = max(r[1] for r in rows) - int(args.skip_last_ms * 1e6)
    rows = [r for r in rows if r[0] <= t_end]
    t0 = max(min(r[0] for r in rows), t_end - int(args.last_ms * 1e6))
    rows = [r for r in rows if r[1] >= t0]
    queues = collections.OrderedDict()
    for s, e, q, n in rows:
        queues.setdefault(q, []).append((s, e, n))
    nbins = int((t_end - t0) / (args.bin * 1e3)) + 1
    print(f"window {1e-6 * (t_end - t0):.1f} ms, {len(rows)} kernels, bin {args.bin} us")
    for q, ks in queues.items():
        busy = sum(e - max(s, t0) for s, e, _ in ks)
        line = [" "] * nbins
        for s, e, n in ks:
            for b in range(int((max(s, t0) - t0) / (args.bin * 1e3)), min(nbins - 1, int((e - t0) / (args.bin * 1e3))) + 1):
                line[b] = klass(n)
        print(f"queue {q:>3}: {len(ks):5d} kernels, busy {1e-6 * busy:8.2f} ms |{''.join(line)}|")
    # longest kernels
    top = sorted(rows, key=lambda r: r[0] - r[1])[:12]
    for s, e, q, n in top:
        print(f"  {1e-6 * (e - s):8.3f} ms  q{q}  start {1e-6 * (s - t0):8.2f}  {n[:90]}")


if __name__ == "__main__":
    main()
