#!/usr/bin/env python
"""cProfile of compress+decompress steps (host-side cost of the model pipeline).
Usage: python tools/model_hostprof.py [bls2017|bmshj2018] [batch]"""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import compression_amd as tfc
from compression_amd import synthetic
which = sys.argv[1] if len(sys.argv) > 1 else "bls2017"
torch.manual_seed(0)
dev = torch.device("cuda", 0)
if which == "bls2017":
    model, batch, hw = tfc.models.BLS2017Model(num_filters=192, compute_dtype=torch.bfloat16), 512, (256, 256)
else:
    model, batch, hw = tfc.models.BMSHJ2018Model(num_filters=192, compute_dtype=torch.bfloat16), 128, (512, 768)
if len(sys.argv) > 2:
    batch = int(sys.argv[2])
model = model.to(dev).init_compression()
base = torch.from_numpy(synthetic.lowpass_images(8, hw[0], hw[1], seed=2)).to(dev)
x = base.repeat((batch + 7) // 8, 1, 1, 1)[:batch].contiguous()
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
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
