#!/usr/bin/env python
"""Turns a rocprofv3 results .db (sqlite, `--kernel-trace --stats`) into the
markdown kernel summary kept under profiles/.  Usage:
    python tools/rocprof_summary.py gpurun_out/prof1 profiles/r01_bench.md "title" """
import glob
import os
import sqlite3
import sys


def main():
    src, dst, title = sys.argv[1], sys.argv[2], sys.argv[3]
    dbs = sorted(glob.glob(os.path.join(src, "**", "*_results.db"), recursive=True))
    assert dbs, f"no *_results.db under {src}"
    lines = [f"# {title}", "", f"source: `{dbs[-1]}` (rocprofv3 --kernel-trace --stats)", "",
             "| kernel | calls | total (us) | average (us) | % |", "|---|---:|---:|---:|---:|"]
    cur = sqlite3.connect(dbs[-1]).cursor()
    for name, calls, total, avg, pct in cur.execute(
            "select name, total_calls, total_duration, average, percentage from top_kernels limit 25"):
        # durations in the view are nanoseconds/1000 = microseconds
        lines.append(f"| `{name[:110]}` | {calls} | {float(total):.1f} | {float(avg):.2f} | {float(pct):.2f} |")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    open(dst, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:12]))


if __name__ == "__main__":
    main()
