#!/usr/bin/env python
"""Third-generation SignalConv2D kernel against the second (TFC_CONV_GEN=2) and a torch fp32 evaluation, and their
times, at layer shapes of the target models.  Usage (GPU box): python tools/conv3_check.py [batch]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("TFC_LIB_PATH"):          # a build variant of the library (tools/conv3_variants.sh)
    import compression_amd._lib as _L
    _L.LIB_PATH = os.environ["TFC_LIB_PATH"]
from compression_amd.layers import conv2d_down, conv2d_up

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
only = sys.argv[2] if len(sys.argv) > 2 else ""          # substring of the case names to run
dev = "cuda"
gen = torch.Generator().manual_seed(3)
cases = [  # name, fn, (H, W), kernel, cin, cout, stride
    ("down 5x5 /2 192 @384x256", conv2d_down, (256, 384), 5, 192, 192, 2),
    ("down 5x5 /2 192 @100x70", conv2d_down, (70, 100), 5, 192, 192, 2),
    ("down 3x3 s1 192 @48x32", conv2d_down, (32, 48), 3, 192, 192, 1),
    ("down 5x5 /2 128 @64x64", conv2d_down, (64, 64), 5, 128, 128, 2),
    ("up 5x5 x2 192 @192x128", conv2d_up, (128, 192), 5, 192, 192, 2),
    ("up 5x5 x2 192 @50x37", conv2d_up, (37, 50), 5, 192, 192, 2),
    ("up 5x5 x2 128 @32x32", conv2d_up, (32, 32), 5, 128, 128, 2),
    ("up 3x3 x1 192 @48x32", conv2d_up, (32, 48), 3, 192, 192, 1),
]


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


if os.environ.get("CONV3_CIN"):      # fixed cost and cost per K step: the big down layer at several input widths
    cases = [(f"down 5x5 /2 cin={ci} @384x256", conv2d_down, (256, 384), 5, ci, 192, 2)
             for ci in map(int, os.environ["CONV3_CIN"].split())] + \
            [(f"up 5x5 x2 cin={ci} @192x128", conv2d_up, (128, 192), 5, ci, 192, 2)
             for ci in map(int, os.environ["CONV3_CIN"].split())]
for name, fn, (H, W), k, ci, co, s in cases:
    if only and only not in name:
        continue
    n = batch if H * W > 20000 else max(batch, 32)
    x = torch.randn(n, H, W, ci, generator=gen).to(torch.bfloat16).to(dev)
    w = (torch.randn(k, k, ci, co, generator=gen) / (k * k * ci) ** 0.5).to(dev)
    bias = torch.randn(co, generator=gen).to(dev)
    out = {}
    ms = {}
    for g in ("3", "2"):
        os.environ["TFC_CONV_GEN"] = "4" if g == "3" else g       # 4: the third generation wherever it is built
        out[g] = fn(x, w, bias, s, "relu")
        ms[g] = timed(lambda: fn(x, w, bias, s, "relu"))
    # torch fp32 on a slice of the batch
    xs = x[:2].float().permute(0, 3, 1, 2)
    wq = w.to(torch.bfloat16).float()
    if fn is conv2d_down:
        ref = F.conv2d(F.pad(xs, (k // 2, k - 1 - k // 2, k // 2, k - 1 - k // 2)), wq.permute(3, 2, 0, 1), bias, stride=s)
        ref = ref[:, :, :out["3"].shape[1], :out["3"].shape[2]]
    else:
        full = F.conv_transpose2d(xs, wq.permute(2, 3, 0, 1), bias, stride=s)
        lo = k // 2
        ref = full[:, :, lo:lo + H * s, lo:lo + W * s]
    ref = torch.relu(ref).permute(0, 2, 3, 1)
    d32 = (out["3"].float() - out["2"].float()).abs().max().item()
    dref3 = (out["3"][:2].float() - ref).abs().max().item()
    dref2 = (out["2"][:2].float() - ref).abs().max().item()
    flops = 2.0 * out["3"].numel() / co * (k * k / (s * s) if fn is conv2d_up else k * k) * ci * co
    print(f"{name:32s} n={n:3d}  gen3 {ms['3']:7.3f} ms ({flops / ms['3'] / 1e9:6.0f} TF)  gen2 {ms['2']:7.3f} ms ({flops / ms['2'] / 1e9:6.0f} TF)"
          f"  |3-2| {d32:.4f}  |3-ref| {dref3:.4f}  |2-ref| {dref2:.4f}  max|ref| {ref.abs().max().item():.2f}")
