#!/usr/bin/env python
"""Per-kernel means of the SQ counters of one rocprofv3 --pmc pass, plus the durations from the kernel
trace of the same run.  Usage: python tools/sq_summary.py <rocprof_dir> <dst-prefix>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import collect, source_stamp  # noqa: E402


def durations(src):
    out = defaultdict(list)
    for f in sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name")
                if name:
                    out[name].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    return out


def main():
    src, dst = sys.argv[1:3]
    ctrs = collect(src)
    dur = durations(src)
    rows = {}
    for name, c in ctrs.items():
        if "SQ_WAVE_CYCLES" not in c:
            continue
        rows[name] = {k: sum(v) / len(v) for k, v in c.items()}
        rows[name]["launches"] = len(c["SQ_WAVE_CYCLES"])
        if name in dur:
            rows[name]["duration_us"] = sum(dur[name]) / len(dur[name])
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    rows["_sources"] = source_stamp()
    json.dump(rows, open(dst + ".json", "w"), indent=1, sort_keys=True)
    del rows["_sources"]
    cols = ["SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS",
            "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_INSTS_VALU"]
    seen = {c for r in rows.values() for c in r}
    cols = [c for c in cols if c in seen] + sorted(c for c in seen if c.startswith("SQ_") and c not in cols)
    lines = ["# SQ counters per kernel (rocprofv3 --pmc, one pass; *_CYCLES / ACTIVE / WAIT in quad-cycles summed over waves)",
             "", "| kernel | launches | duration us | " + " | ".join(cols) + " |", "|---|---:|---:|" + "---:|" * len(cols)]
    for name in sorted(rows, key=lambda k: -rows[k].get("SQ_WAVE_CYCLES", 0))[:12]:
        r = rows[name]
        lines.append(f"| `{name[:70]}` | {r['launches']} | {r.get('duration_us', float('nan')):.1f} | "
                     + " | ".join(f"{r.get(c, float('nan')):.4g}" for c in cols) + " |")
    open(dst + ".md", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
