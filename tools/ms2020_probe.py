"""Timing of ms2020 compress + decompress (full-size model, random weights): ms per image batch, and the split
between the slice loop's coder calls and everything else."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import compression_amd as tfc
from compression_amd import synthetic, _lib
import ctypes as C

def q(name):
    ms, n = C.c_double(), C.c_int64()
    _lib.lib().tfc_profile_query(name.encode(), C.byref(ms), C.byref(n))
    return ms.value, n.value

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
model = tfc.models.MS2020Model(compute_dtype=torch.bfloat16).cuda().init_compression()
x = torch.from_numpy(synthetic.lowpass_images(8, 512, 768, seed=3)).cuda().repeat((B + 7) // 8, 1, 1, 1)[:B].contiguous()
for _ in range(2):
    out = model.compress(x); rec = model.decompress(*out)
torch.cuda.synchronize()
_lib.lib().tfc_profile_enable(1)
t0 = time.perf_counter(); out = model.compress(x); torch.cuda.synchronize(); t1 = time.perf_counter()
rec = model.decompress(*out); torch.cuda.synchronize(); t2 = time.perf_counter()
k = {n: q(n) for n in ("enc_kernel", "dec_kernel", "conv2d", "gdn_forward")}
_lib.lib().tfc_profile_enable(0)
nbytes = sum(len(bytes(s)) for arr in out[3:] for s in arr)
print(f"ms2020, {B} images of 768x512 bf16: compress {1e3 * (t1 - t0):.1f} ms, decompress {1e3 * (t2 - t1):.1f} ms, "
      f"{8 * nbytes / (B * 512 * 768):.3f} bpp; kernels (ms, launches): " + ", ".join(f"{n} {v[0]:.1f}/{v[1]}" for n, v in k.items()))
