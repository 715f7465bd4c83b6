#!/usr/bin/env python
"""Round 6: every SignalConv2D of a bmshj2018 step in float32 at `batch` images of 768x512 — ms per layer (TFC_CONV_F32 as set)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from compression_amd.layers import conv2d_down, conv2d_up
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
C, H, W = 192, 512, 768
gen = torch.Generator().manual_seed(5)
layers = [("ana 5x5 3->192 /2", conv2d_down, (H, W, 3), 5, 3, C, 2), ("ana 5x5 /2 @384x256", conv2d_down, (H // 2, W // 2, C), 5, C, C, 2),
          ("ana 5x5 /2 @192x128", conv2d_down, (H // 4, W // 4, C), 5, C, C, 2), ("ana 5x5 /2 @96x64", conv2d_down, (H // 8, W // 8, C), 5, C, C, 2),
          ("hyp 3x3 s1 @48x32", conv2d_down, (H // 16, W // 16, C), 3, C, C, 1), ("hyp 5x5 /2 @48x32", conv2d_down, (H // 16, W // 16, C), 5, C, C, 2),
          ("syn 5x5 x2 @48x32", conv2d_up, (H // 16, W // 16, C), 5, C, C, 2), ("syn 5x5 x2 @96x64", conv2d_up, (H // 8, W // 8, C), 5, C, C, 2),
          ("syn 5x5 x2 @192x128", conv2d_up, (H // 4, W // 4, C), 5, C, C, 2), ("syn 5x5 192->3 x2 @384x256", conv2d_up, (H // 2, W // 2, C), 5, C, 3, 2)]
tot = 0.0
for name, fn, shp, k, ci, co, s in layers:
    x = torch.randn((batch,) + shp, generator=gen).cuda()
    w = (torch.randn(k, k, ci, co, generator=gen) / (k * k * ci) ** 0.5).cuda()
    b = torch.zeros(co).cuda()
    fn(x, w, b, s); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(2): fn(x, w, b, s)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 2
    tot += ms
    print(f"{os.environ.get('TFC_CONV_F32', 'split'):7s} {name:30s} n={batch} {ms:8.3f} ms", flush=True)
    del x
print("sum", round(tot, 2))
