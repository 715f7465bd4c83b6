#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), csv output.
Writes <dst>.json (raw per-kernel averages, counter units as reported) and <dst>.md.
Usage: python tools/pmc_summary.py <fetch_dir> <write_dir> <dst-prefix>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def source_stamp():
    """git blob hashes of the kernel sources the profile is taken on (bench.py compares them with the
    tree it runs in and drops counter figures of other code)."""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for rel in ("compression_amd/csrc/range_coder.hip", "compression_amd/csrc/range_lanes.h", "compression_amd/csrc/range_pipe.h",
                "compression_amd/csrc/range_encoder_fast.h", "compression_amd/csrc/range_decoder_fast.h",
                "compression_amd/csrc/gdn.hip", "compression_amd/csrc/gdn_common.h",
                "compression_amd/csrc/gdn_backward.hip"):
        try:
            data = open(os.path.join(root, rel), "rb").read()
        except OSError:
            continue
        out[rel] = hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()
    # coding steps per lane-kernel launch of the profiled command (per-launch counters only compare with a run
    # that launches the same number): exported by the profile script
    if os.environ.get("TFC_PROFILE_STEPS_PER_LAUNCH"):
        out["_steps_per_launch"] = int(os.environ["TFC_PROFILE_STEPS_PER_LAUNCH"])
    return out


def collect(src):
    out = defaultdict(lambda: defaultdict(list))
    files = sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True))
    assert files, f"no *counter_collection.csv under {src}"
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name") or row.get("kernel_name")
                ctr = row.get("Counter_Name") or row.get("counter_name")
                val = row.get("Counter_Value") or row.get("counter_value")
                if name and ctr and val not in (None, ""):
                    out[name][ctr].append(float(val))
    return out


def main():
    fetch_dir, write_dir, dst = sys.argv[1:4]
    merged = defaultdict(dict)
    for src in (fetch_dir, write_dir):
        for name, ctrs in collect(src).items():
            for ctr, vals in ctrs.items():
                merged[name][ctr] = {"launches": len(vals), "mean": sum(vals) / len(vals),
                                     "min": min(vals), "max": max(vals)}
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    merged["_sources"] = source_stamp()
    json.dump(merged, open(dst + ".json", "w"), indent=1, sort_keys=True)
    del merged["_sources"]
    lines = ["# HBM traffic counters per kernel (rocprofv3 --pmc, separate FETCH_SIZE / WRITE_SIZE passes)", "",
             "Counter values as reported (KiB per dispatch); corrections are applied in bench.py / DESIGN.md.", "",
             "| kernel | launches | FETCH_SIZE mean | WRITE_SIZE mean |", "|---|---:|---:|---:|"]
    for name in sorted(merged, key=lambda k: -merged[k].get("FETCH_SIZE", {}).get("mean", 0)):
        f = merged[name].get("FETCH_SIZE", {})
        w = merged[name].get("WRITE_SIZE", {})
        lines.append(f"| `{name[:100]}` | {f.get('launches', w.get('launches', 0))} | "
                     f"{f.get('mean', float('nan')):.1f} | {w.get('mean', float('nan')):.1f} |")
    open(dst + ".md", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:30]))


if __name__ == "__main__":
    main()
