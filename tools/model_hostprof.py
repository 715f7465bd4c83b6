#!/usr/bin/env python
"""cProfile of bls2017 compress+decompress steps (host-side cost of the model pipeline)."""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import compression_amd as tfc
from compression_amd import synthetic
torch.manual_seed(0)
dev = torch.device("cuda", 0)
model = tfc.models.BLS2017Model(num_filters=192, compute_dtype=torch.bfloat16).to(dev).init_compression()
base = torch.from_numpy(synthetic.lowpass_images(8, 256, 256, seed=2)).to(dev)
x = base.repeat(64, 1, 1, 1).contiguous()
def step():
    out = model.compress(x)
    return model.decompress(*out)
for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    step()
torch.cuda.synchronize()
pr.disable()
print("ms/step", (time.perf_counter() - t0) / 3 * 1e3)
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
