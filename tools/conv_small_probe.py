#!/usr/bin/env python
"""The two image-side layers of bmshj2018 at the C4 shape (128 x 768x512): time and bytes.  Usage (GPU box):
python tools/conv_small_probe.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from compression_amd.layers import conv2d_down, conv2d_up

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
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


x = torch.randn(batch, 256, 384, 192, generator=gen).to(torch.bfloat16).cuda()
w = (torch.randn(5, 5, 192, 3, generator=gen) / 70).cuda()
bias = torch.randn(3, generator=gen).cuda()
ms = timed(lambda: conv2d_up(x, w, bias, 2))
gb = (x.numel() * 2 + batch * 512 * 768 * 3 * 2) / 1e9
print(f"synthesis 5x5 192->3 x2 @384x256 n={batch}: {ms:.3f} ms, {gb:.2f} GB in+out = {gb / ms:.2f} TB/s")
xi = torch.rand(batch, 512, 768, 3, generator=gen).to(torch.bfloat16).cuda()
w0 = (torch.randn(5, 5, 3, 192, generator=gen) / 9).cuda()
b0 = torch.randn(192, generator=gen).cuda()
ms = timed(lambda: conv2d_down(xi, w0, b0, 2))
gb = (xi.numel() * 2 + batch * 256 * 384 * 192 * 2) / 1e9
print(f"analysis 5x5 3->192 /2 @768x512 n={batch}: {ms:.3f} ms, {gb:.2f} GB in+out = {gb / ms:.2f} TB/s")
