#!/usr/bin/env python
"""Does a model sub-benchmark depend on what ran before it in the same process?  Runs bench.model_bench for the
workloads named, in order.  Usage (GPU box): python tools/bench_sequence_probe.py bls2017 bmshj2018"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from compression_amd import pipeline

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
for name in sys.argv[1:]:
    torch.cuda.empty_cache()
    if name == "trim":
        print("trim", pipeline.empty_cache(), "bytes; cached", pipeline.cached_bytes())
        continue
    r = bench.model_bench(name, "bf16", dev, steps=32, warmup=2, depth=0, coder_cus=0, cpu=False)
    print(name, r["value"], r["ms_per_step"], "library cache", pipeline.cached_bytes() >> 20, "MiB")
