#!/usr/bin/env python
"""Kernel-time probe for SignalConv2D: TFLOP/s on the layer shapes of the two models."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from compression_amd import _lib
from compression_amd.layers import conv2d_down, conv2d_up, gdn_forward


def q(name):
    ms, n = C.c_double(), C.c_int64()
    _lib.lib().tfc_profile_query(name.encode(), C.byref(ms), C.byref(n))
    return ms.value / max(n.value, 1)


def run(label, fn, x, k, bias, stride, up, dtype):
    x = x.to(dtype)
    for r in range(4):
        if r == 1:
            _lib.lib().tfc_profile_enable(1)
        y = fn(x, k, bias, stride)
    torch.cuda.synchronize()
    ms = q("conv2d")
    _lib.lib().tfc_profile_enable(0)
    n, h, w, cin = x.shape
    kh, kw, _, cout = k.shape
    m_out = y.shape[0] * y.shape[1] * y.shape[2]
    flops = 2.0 * m_out * kh * kw * cin * cout / (stride * stride if up else 1)
    print(f"{label:44s} {str(tuple(x.shape)):24s} -> {str(tuple(y.shape)):24s} {ms:8.3f} ms  {flops / ms / 1e9:8.1f} TFLOP/s ({dtype})")


if __name__ == "__main__":
    dev = torch.device("cuda")
    C_ = 192
    for dtype in (torch.bfloat16, torch.float32):
        B = int(os.environ.get("BATCH", 64)) if dtype == torch.bfloat16 else int(os.environ.get("BATCH_F32", 8))
        img = torch.rand(B, 256, 256, 3, device=dev)
        run("bls2017 analysis L0 9x9 3->C /4", conv2d_down, img, torch.randn(9, 9, 3, C_) / 16, torch.zeros(C_), 4, False, dtype)
        x1 = torch.randn(B, 64, 64, C_, device=dev)
        run("analysis 5x5 C->C /2 (64x64)", conv2d_down, x1, torch.randn(5, 5, C_, C_) / 70, torch.zeros(C_), 2, False, dtype)
        x2 = torch.randn(B, 32, 32, C_, device=dev)
        run("analysis 5x5 C->C /2 (32x32)", conv2d_down, x2, torch.randn(5, 5, C_, C_) / 70, None, 2, False, dtype)
        y = torch.randn(B, 16, 16, C_, device=dev)
        run("synthesis 5x5 C->C x2 (16x16)", conv2d_up, y, torch.randn(5, 5, C_, C_) / 70, torch.zeros(C_), 2, True, dtype)
        y2 = torch.randn(B, 32, 32, C_, device=dev)
        run("synthesis 5x5 C->C x2 (32x32)", conv2d_up, y2, torch.randn(5, 5, C_, C_) / 70, torch.zeros(C_), 2, True, dtype)
        y3 = torch.randn(B, 64, 64, C_, device=dev)
        run("bls2017 synthesis 9x9 C->3 x4", conv2d_up, y3, torch.randn(9, 9, C_, 3) / 120, torch.zeros(3), 4, True, dtype)
        run("hyper 3x3 C->C s1 (32x32)", conv2d_down, x2, torch.randn(3, 3, C_, C_) / 40, torch.zeros(C_), 1, False, dtype)
