"""Times tfc.stochastic_round on the bench workload's latent tensor (512 x 16 x 16 x 192 elements).
Run on the GPU box:  python tools/stochastic_round_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compression_amd as tfc  # noqa: E402

for dtype in (torch.float32, torch.bfloat16, torch.float16):
    for n in (512 * 16 * 16 * 192, 1 << 20, 1 << 16):
        x = (torch.rand(n, device="cuda") * 200 - 100).to(dtype)
        for _ in range(3):
            tfc.stochastic_round(x, 0.5, (1, 2))
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        reps = 20
        for _ in range(reps):
            tfc.stochastic_round(x, 0.5, (1, 2))
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / reps
        nbytes = n * (x.element_size() + 4)
        print(f"{str(dtype):16s} n={n:9d}  {us:9.1f} us  {nbytes / us / 1e3:8.1f} GB/s of tensor traffic")
