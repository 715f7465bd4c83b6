"""The elementwise passes around the transforms at the C4 shape (128 x 768x512x3 images; 128 x 48x32x192 indexes):
HIP-event time and bytes in + out per second.  Usage (GPU box): python tools/elementwise_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from compression_amd.layers import functional
x8 = torch.randint(0, 256, (128, 768, 512, 3), dtype=torch.uint8, device="cuda")
u = functional.image_to_unit(x8, torch.bfloat16)
idx = (torch.rand(128, 48, 32, 192, device="cuda") * 70 - 3).to(torch.bfloat16)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
from compression_amd import _lib
out = torch.empty(idx.shape, dtype=torch.int32, device="cuda")
def prep():
    _lib.check(_lib.lib().tfc_index_prepare(idx.data_ptr(), 1, out.data_ptr(), idx.numel(), 64, _lib.stream_ptr()))
for name, fn, nbytes in (("image_to_unit", lambda: functional.image_to_unit(x8, torch.bfloat16), x8.numel() * 3),
                         ("unit_to_image", lambda: functional.unit_to_image(u), x8.numel() * 3),
                         ("index_prepare", prep, idx.numel() * 6)):
    ms = t(fn)
    print(f"{name}: {ms * 1e3:.0f} us, {nbytes / ms / 1e6:.0f} GB/s")
