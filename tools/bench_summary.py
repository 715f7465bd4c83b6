#!/usr/bin/env python
"""Prints the figures of a bench.py JSON line that a round's notes quote (python tools/bench_summary.py LOGFILE)."""
import json
import sys

line = [l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1]
d = json.loads(line)
print("value", d["value"], "ms/step", d["ms_per_step"], "x cpu", d.get("speedup_vs_cpu_baseline"), "cpu", d.get("cpu_baseline", {}).get("value"),
      "slots", d.get("cpu_baseline", {}).get("slots_compared"), d.get("cpu_baseline", {}).get("slots_identical"))
print(d["config"].get("single_batch"))
print("stages", json.dumps(d["kernels_ms_in_flight"]["stages"]))
for r in d.get("saturation", {}).get("points", []):
    print("sat", r["batches_per_launch"], r["mpixels_s"], "ms/group", r["ms_per_group"], "enc", r["encode_call_ms"], "dec", r["decode_call_ms"],
          r["kernels_ms"], "launches", r["launches_per_direction"], r["dominant_kernel"], r["dominant_kernel_algorithmic_gbs"],
          "path GB/s", r["path_algorithmic_gbs"], "x", r.get("speedup_vs_cpu_baseline"))
g = d.get("gdn_fwd")
if g:
    print("gdn", {k: g[k] for k in ("kernel_ms", "achieved", "frac", "warm", "kernel_ms_single_launch_events") if k in g})
    print("copy", g.get("copy_reference"))
    for k, v in g["variants"].items():
        if k != "note":
            print(" ", k, "fwd", v["forward"]["kernel_ms"], v["forward"]["frac"], "warm", v["forward"]["warm_kernel_ms"],
                  "bwd", v["backward"]["kernel_ms"], v["backward"]["frac"], v["backward"]["passes_ms"])
print("roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["traffic"])
for k, m in d.get("models", {}).items():
    print(k, m["value"], "ms", m["ms_per_step"], "lone", m["lone_step"]["ms_per_step"], m["lone_step"]["kernels_ms"], "TF/s", m["roofline"]["achieved"])
for r in d.get("conv", {}).get("layers", []):
    print(r)
if "escapes" in d:
    print("escapes", d["escapes"]["0.01"]["value"], d["escapes"]["escape_free"]["value"])
if "training" in d:
    t = d["training"]
    print("training", t["ms_per_step"], t["kernels_ms"], t["wgrad_5x5_s2_192_192_at_16x384x256"])

