#!/usr/bin/env python
"""Fused training-time bottleneck (csrc/factorized_bits.hip) against the op-by-op torch
evaluation, on the C2 latent tensor [512, 16, 16, 192]."""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import compression_amd as tfc
from compression_amd import _lib
from compression_amd.ops import bottleneck_ops


def q(name):
    ms, n = C.c_double(), C.c_int64()
    _lib.lib().tfc_profile_query(name.encode(), C.byref(ms), C.byref(n))
    return ms.value / max(n.value, 1)


torch.manual_seed(0)
for dtype in (torch.bfloat16, torch.float32):
    prior = tfc.NoisyDeepFactorized(batch_shape=(192,)).cuda()
    y = (2 * torch.randn(512, 16, 16, 192, device="cuda")).to(dtype).requires_grad_(True)
    noise = (torch.rand_like(y) - 0.5)

    def fused():
        y.grad = None
        y_hat, bits = bottleneck_ops.factorized_bits(y, prior.base, 3, noise)
        bits.sum().backward()

    def reference():
        y.grad = None
        y_hat = y + noise
        bits = prior.log_prob(y_hat.float()).sum(dim=(1, 2, 3)) / -float(np.log(2.0))
        bits.sum().backward()

    for fn in (fused, reference):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        _lib.lib().tfc_profile_enable(1)
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5 * 1e3
        extra = ""
        if fn is fused:
            f, b = q("factorized_forward"), q("factorized_backward")
            n = y.numel()
            es = y.element_size()
            extra = (f"  kernels: forward {f:.3f} ms ({3 * n * es / f / 1e6:.0f} GB/s of y, u, y_hat), "
                     f"backward {b:.3f} ms ({2 * n * es / b / 1e6:.0f} GB/s of y_hat, dy)")
        _lib.lib().tfc_profile_enable(0)
        print(f"{str(dtype):15s} {fn.__name__:10s} fwd+bwd {dt:8.3f} ms per step (wall){extra}")
