#!/usr/bin/env python
"""Host-side timing of the model software pipeline: how long each stage's enqueue call takes and when, to find
where the enqueuing thread blocks.  Usage (GPU box): python tools/pipeline_host_log.py [bmshj2018|bls2017] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from compression_amd import pipeline

workload = sys.argv[1] if len(sys.argv) > 1 else "bmshj2018"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda", 0)
model, x, batch, hw, _ = bench.make_model(workload, torch.bfloat16, dev, 0, 0)
part = pipeline.CoderPartition(coder_cus=32 if workload == "bmshj2018" else 128, depth=1, device=dev, mode="masked")
side = torch.cuda.Stream(device=dev)
with torch.cuda.stream(side):
    bench.run_model_pipeline(model, x, 3, pipeline.SoftwarePipeline(part.lanes[0]))
    torch.cuda.synchronize()
    sp = pipeline.SoftwarePipeline(part.lanes[0])
    sp.host_log = []
    t0 = time.perf_counter()
    import cProfile, pstats
    prof = cProfile.Profile()
    prof.enable()
    el, _ = bench.run_model_pipeline(model, x, steps, sp)
    prof.disable()
    pstats.Stats(prof).sort_stats("tottime").print_stats(18)
print(f"{workload}: {1e3 * el / steps:.2f} ms per step")
for name, a, b in sp.host_log:
    print(f"{1e3 * (a - t0):9.2f} {1e3 * (b - a):8.2f} ms  {name}")
