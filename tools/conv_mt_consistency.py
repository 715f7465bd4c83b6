"""Does a SignalConv2D output depend on the batch it is computed in (1 vs 2 pixel tiles per wave)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from compression_amd.layers import conv2d_down, conv2d_up, gdn_forward
torch.manual_seed(1)
dev = torch.device("cuda")
C = 192
for label, fn, shape, k, stride in (
        ("down 5x5 /2", conv2d_down, (128, 64, 96, C), torch.randn(5, 5, C, C) / 70, 2),
        ("up 5x5 x2", conv2d_up, (128, 32, 48, C), torch.randn(5, 5, C, C) / 70, 2),
        ("down 3x3 s1", conv2d_down, (128, 32, 48, C), torch.randn(3, 3, C, C) / 40, 1),
        ("up 5x5 x2 C->3", conv2d_up, (32, 128, 192, C), torch.randn(5, 5, C, 3) / 70, 2)):
    x = torch.randn(*shape, device=dev).bfloat16()
    b = torch.randn(k.shape[-1]) / 10
    full = fn(x, k, b, stride)
    parts = torch.cat([fn(x[i:i + 16].contiguous(), k, b, stride) for i in range(0, shape[0], 16)])
    print(label, "full vs 16-image slices: equal", bool(torch.equal(full, parts)), "max abs diff",
          float((full.float() - parts.float()).abs().max()))
x = torch.randn(128 * 32 * 48, C, device=dev).bfloat16()
beta, gamma = 1 + 0.1 * torch.rand(C), 0.1 * torch.eye(C) + 0.01 * torch.rand(C, C)
y = gdn_forward(x, beta, gamma)
yp = torch.cat([gdn_forward(x[i:i + 4096].contiguous(), beta, gamma) for i in range(0, x.shape[0], 4096)])
print("gdn full vs slices equal", bool(torch.equal(y, yp)))
