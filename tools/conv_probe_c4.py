#!/usr/bin/env python
"""SignalConv2D kernel time per layer at the C4 scale: bmshj2018 on [batch] images of 768x512, bf16."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conv_probe import run
from compression_amd.layers import conv2d_down, conv2d_up

if __name__ == "__main__":
    dev = torch.device("cuda")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    C_, dt = 192, torch.bfloat16
    H, W = 512, 768
    k5 = lambda ci, co: torch.randn(5, 5, ci, co) / 70
    only = os.environ.get("LAYERS")           # e.g. LAYERS=L1,S2: substrings of the labels to run
    _run = run
    def run(label, *a):
        if only is None or any(tag in label for tag in only.split(",")):
            _run(label, *a)
    run("analysis L0 5x5 3->C /2", conv2d_down, torch.rand(B, H, W, 3, device=dev), k5(3, C_), torch.zeros(C_), 2, False, dt)
    run("analysis L1 5x5 C->C /2", conv2d_down, torch.randn(B, H // 2, W // 2, C_, device=dev, dtype=dt), k5(C_, C_), torch.zeros(C_), 2, False, dt)
    run("analysis L2 5x5 C->C /2", conv2d_down, torch.randn(B, H // 4, W // 4, C_, device=dev, dtype=dt), k5(C_, C_), torch.zeros(C_), 2, False, dt)
    run("analysis L3 5x5 C->C /2", conv2d_down, torch.randn(B, H // 8, W // 8, C_, device=dev, dtype=dt), k5(C_, C_), None, 2, False, dt)
    run("hyper-a 3x3 C->C s1", conv2d_down, torch.randn(B, H // 16, W // 16, C_, device=dev, dtype=dt), torch.randn(3, 3, C_, C_) / 40, torch.zeros(C_), 1, False, dt)
    run("hyper-a 5x5 C->C /2", conv2d_down, torch.randn(B, H // 16, W // 16, C_, device=dev, dtype=dt), k5(C_, C_), torch.zeros(C_), 2, False, dt)
    run("hyper-s 5x5 C->C x2", conv2d_up, torch.randn(B, H // 64, W // 64, C_, device=dev, dtype=dt), k5(C_, C_), torch.zeros(C_), 2, True, dt)
    run("synthesis S0 5x5 C->C x2", conv2d_up, torch.randn(B, H // 16, W // 16, C_, device=dev, dtype=dt), k5(C_, C_), torch.zeros(C_), 2, True, dt)
    run("synthesis S1 5x5 C->C x2", conv2d_up, torch.randn(B, H // 8, W // 8, C_, device=dev, dtype=dt), k5(C_, C_), torch.zeros(C_), 2, True, dt)
    run("synthesis S2 5x5 C->C x2", conv2d_up, torch.randn(B, H // 4, W // 4, C_, device=dev, dtype=dt), k5(C_, C_), torch.zeros(C_), 2, True, dt)
    run("synthesis S3 5x5 C->3 x2", conv2d_up, torch.randn(B, H // 2, W // 2, C_, device=dev, dtype=dt), k5(C_, 3), torch.zeros(3), 2, True, dt)
