#!/usr/bin/env python
"""One bls2017 training step (forward + backward + Adam) on the HIP kernels: wall time and the
per-kernel split.  python tools/train_probe.py [batch] (on a GPU box)"""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from compression_amd import _lib, models, synthetic

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.manual_seed(0)
for dtype in (torch.bfloat16, torch.float32):
    model = models.BLS2017Model(lmbda=0.01, num_filters=192, compute_dtype=dtype).cuda()
    x = torch.from_numpy(synthetic.lowpass_images(8, 256, 256, seed=3)).cuda().repeat((batch + 7) // 8, 1, 1, 1)[:batch]
    model(x)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)

    def step():
        opt.zero_grad()
        loss, bpp, mse = model(x, training=True)
        loss.backward()
        opt.step()
        return float(loss.detach())

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    _lib.lib().tfc_profile_enable(1)
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    parts = []
    for name in ("conv2d", "conv2d_wgrad", "gdn_forward", "gdn_backward_fused", "gdn_backward_t", "gdn_backward_dx",
                 "gdn_backward_params", "factorized_forward", "factorized_backward"):
        ms, cnt = C.c_double(), C.c_int64()
        _lib.lib().tfc_profile_query(name.encode(), C.byref(ms), C.byref(cnt))
        if cnt.value:
            parts.append(f"{name} {ms.value / n:.2f} ms ({cnt.value // n}x)")
    _lib.lib().tfc_profile_enable(0)
    print(f"bls2017 train step, batch {batch} x 256x256, {dtype}: {dt:.1f} ms/step "
          f"({batch * 65536 / 1e6 / (dt / 1e3):.1f} Mpixels/s), loss {loss:.3f}\n   " + ", ".join(parts))
