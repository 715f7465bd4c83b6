#!/usr/bin/env python
"""Round 6: float32 SignalConv2D at the C4 layer shapes — the float32 MFMA kernel (TFC_CONV_F32=native) against the
bf16 x 6 split on the bfloat16 kernels (default): time, TFLOP/s of the float32 problem, and the largest difference between
the two and to a float64 torch evaluation on a slice.  Usage (GPU box): TFC_CONV_F32=native|split python tools/r06_f32_probe.py [batch]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from compression_amd.layers import conv2d_down, conv2d_up

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
gen = torch.Generator().manual_seed(3)
cases = [("down 5x5 /2 192 @384x256", conv2d_down, (256, 384), 5, 192, 192, 2),
         ("down 5x5 /2 192 @96x64", conv2d_down, (64, 96), 5, 192, 192, 2),
         ("down 3x3 s1 192 @48x32", conv2d_down, (32, 48), 3, 192, 192, 1),
         ("up 5x5 x2 192 @192x128", conv2d_up, (128, 192), 5, 192, 192, 2),
         ("up 5x5 x2 192 @48x32", conv2d_up, (32, 48), 5, 192, 192, 2)]
for name, fn, (H, W), k, ci, co, s in cases:
    x = torch.randn(batch, H, W, ci, generator=gen).cuda()
    w = (torch.randn(k, k, ci, co, generator=gen) / (k * k * ci) ** 0.5).cuda()
    b = torch.randn(co, generator=gen).cuda()
    y = fn(x, w, b, s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        fn(x, w, b, s)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / 3
    xs = x[:1].double().permute(0, 3, 1, 2).cpu()
    wd = w.double().cpu()
    if fn is conv2d_down:
        ref = F.conv2d(F.pad(xs, (k // 2, k - 1 - k // 2, k // 2, k - 1 - k // 2)), wd.permute(3, 2, 0, 1), b.double().cpu(), stride=s)
        ref = ref[:, :, :y.shape[1], :y.shape[2]]
    else:
        full = F.conv_transpose2d(xs, wd.permute(2, 3, 0, 1), b.double().cpu(), stride=s)
        ref = full[:, :, k // 2:k // 2 + H * s, k // 2:k // 2 + W * s]
    ref = ref.permute(0, 2, 3, 1)
    err = (y[:1].double().cpu() - ref).abs().max().item()
    flops = 2.0 * y.numel() / co * (k * k / (s * s) if fn is conv2d_up else k * k) * ci * co
    print(f"{os.environ.get('TFC_CONV_F32', 'split'):7s} {name:28s} n={batch} {ms:8.3f} ms {flops / ms / 1e9:7.1f} TFLOP/s  max|y - float64| {err:.2e}  max|ref| {ref.abs().max().item():.2f}", flush=True)
    del x, y
