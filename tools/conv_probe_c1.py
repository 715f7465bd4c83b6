#!/usr/bin/env python
"""SignalConv2D layers of bls2017 at the C1 shape (512 x 256x256, 128 filters, bf16): ms per layer.
Usage (GPU box): python tools/conv_probe_c1.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from compression_amd.layers import conv2d_down, conv2d_up

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
gen = torch.Generator().manual_seed(1)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


F = int(sys.argv[2]) if len(sys.argv) > 2 else 128
layers = [(f"analysis 9x9 3->{F} /4 @256", conv2d_down, 256, 9, 3, F, 4),
          ("analysis 5x5 F->F /2 @64", conv2d_down, 64, 5, F, F, 2),
          ("analysis 5x5 F->F /2 @32", conv2d_down, 32, 5, F, F, 2),
          ("synthesis 5x5 F->F x2 @16", conv2d_up, 16, 5, F, F, 2),
          ("synthesis 5x5 F->F x2 @32", conv2d_up, 32, 5, F, F, 2),
          ("synthesis 9x9 F->3 x4 @64", conv2d_up, 64, 9, F, 3, 4)]
total = 0.0
for name, fn, hw, k, ci, co, s in layers:
    x = (torch.rand if ci == 3 else torch.randn)(batch, hw, hw, ci, generator=gen).to(torch.bfloat16).cuda()
    w = (torch.randn(k, k, ci, co, generator=gen) / (k * ci ** 0.5)).cuda()
    bias = torch.randn(co, generator=gen).cuda()
    ms = timed(lambda: fn(x, w, bias, s))
    total += ms
    out = batch * (hw * s if fn is conv2d_up else hw // s) ** 2 * co
    gb = (x.numel() + out) * 2 / 1e9
    print(f"{name:34s} {ms:7.3f} ms   in+out {gb:5.2f} GB = {gb / ms:5.2f} TB/s")
print(f"sum {total:.3f} ms")
