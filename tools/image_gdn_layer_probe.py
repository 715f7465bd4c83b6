"""bmshj2018's first analysis layer at the C4 shape (128 x 768x512x3 -> 384x256x192, GDN behind it): one kernel
(conv_image_gdn_kernel) against conv_image_kernel + the GDN kernel, lone, HIP-event time over 20 launches."""
import os
import torch
from compression_amd import layers

torch.manual_seed(0)
x = torch.rand(128, 768, 512, 3, device="cuda").mul(255).to(torch.bfloat16)
for fused in (True, False, True, False):
    gdn = layers.GDN()
    conv = layers.SignalConv2D(192, (5, 5), corr=True, strides_down=2, padding="same_zeros", in_channels=3, use_bias=True,
                               activation=gdn).cuda()
    conv.fuse_gdn_image = fused
    with torch.no_grad():
        for _ in range(3):
            y = conv(x)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            y = conv(x)
        b.record()
        torch.cuda.synchronize()
    out_gb = y.numel() * 2 / 1e9
    ms = a.elapsed_time(b) / 20
    print(f"fused={fused}: {ms:.3f} ms per layer  (output {out_gb:.2f} GB -> {out_gb / ms * 1e3:.0f} GB/s of output alone)")

# the memory system's side of it: writing (fill) and copying a tensor of the layer's output size
y = torch.empty(128, 384, 256, 192, device="cuda", dtype=torch.bfloat16)
z = torch.empty_like(y)
for name, fn, gb in (("fill", lambda: y.fill_(1.0), y.numel() * 2 / 1e9), ("copy", lambda: z.copy_(y), y.numel() * 4 / 1e9)):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        fn()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(f"{name} of {y.numel() * 2 / 1e9:.2f} GB: {ms:.3f} ms = {gb / ms * 1e3:.0f} GB/s")
