#!/usr/bin/env python
"""On the GPU box: time of the fused factorized-bits backward kernel per element, for the MLP shapes the
library builds, at a channel count that takes the <= 256-thread build (192) and one that takes the 512-thread
build (320).  Usage: python tools/fb_bwd_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compression_amd as tfc
from compression_amd.ops import bottleneck_ops

for nf in ((3, 3), (3, 3, 3), (5, 5)):
    for C in (192, 320):
        prior = tfc.NoisyDeepFactorized(batch_shape=(C,), num_filters=nf).cuda()
        y = torch.randn(8, 64, 64, C, device="cuda", requires_grad=True)
        noise = torch.rand_like(y) - 0.5
        times = []
        for it in range(6):
            y.grad = None
            _, bits = bottleneck_ops.factorized_bits(y, prior.base, 3, noise)
            loss = bits.sum()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            loss.backward()
            b.record()
            torch.cuda.synchronize()
            times.append(a.elapsed_time(b))
        t = sorted(times[1:])[len(times[1:]) // 2]
        print(f"num_filters={nf} C={C}: backward {t:.3f} ms, {t * 1e6 / y.numel():.3f} ns/element", flush=True)
