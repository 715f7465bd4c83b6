#!/usr/bin/env python
"""Where the wave-per-stream decoder's cycles go in index mode (bmshj2018's conditional Gaussian): 128 streams x
65536 symbols, the model's own 64 scale tables, synthetic index / symbol fields:
  narrow   every symbol from a row of <= 64 symbols       wide   every symbol from a wide row
  mix      the bench's calibrated index histogram          mix+e  ... with 0.6 % of the symbols escaping
Prints kernel milliseconds (library HIP-event timers) and cycles per symbol at 2.4 GHz.
python tools/index_decode_probe.py   (on a GPU box)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import compression_amd as tfc
from compression_amd import _lib, synthetic

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = tfc.models.BMSHJ2018Model(num_filters=192, compute_dtype=torch.bfloat16).to(dev).init_compression()
em = model.entropy_model
lookup = em.cdf.cpu().numpy()
rows = synthetic.lookup_rows(lookup)
width = np.array([len(c) - 2 for _, c in rows])
print("plain symbols per table:", width.tolist())
S, E = 128, 65536
rng = np.random.default_rng(0)
hist = np.exp(-0.5 * ((np.arange(64) - 20.0) / 12.0) ** 2)
hist[0] += hist[:1].sum() * 2
hist /= hist.sum()


def make(kind):
    if kind == "narrow":
        index = rng.integers(5, 30, (S, E))
    elif kind == "wide":
        index = rng.integers(40, 60, (S, E))
    else:
        index = rng.choice(64, size=(S, E), p=hist)
    index = index.astype(np.int32)
    value = np.zeros((S, E), np.int32)
    for t, (sp, cdf) in enumerate(rows):
        m = index == t
        u = rng.integers(0, 1 << 12, int(m.sum()))
        value[m] = np.minimum(np.searchsorted(cdf, u, side="right") - 1, len(cdf) - 3)
    if kind.startswith("mix+e"):
        esc = rng.random((S, E)) < float(kind[5:] or 0.006)
        value[esc] = width[index[esc]] + rng.geometric(0.2, int(esc.sum()))
    return torch.from_numpy(index).to(dev), torch.from_numpy(value).to(dev)


lt = em.cdf
for kind in (sys.argv[1:] or ("narrow", "wide", "mix", "mix+e")):
    it, vt = make(kind)
    res = []
    for rep in range(3):
        _lib.lib().tfc_profile_enable(1)
        h = tfc.create_range_encoder([S], lt, mode="latency")
        h = tfc.entropy_encode_index(h, it, vt)
        blob, off = tfc.gen_ops._finalize_device(h)
        d = tfc.create_range_decoder((blob, off, (S,)), lt, mode="latency")
        d, dec = tfc.entropy_decode_index(d, it, [E], torch.int32)
        ok = tfc.entropy_decode_finalize(d)
        torch.cuda.synchronize()
        res.append((bench.profile_query("enc_kernel")[0], bench.profile_query("dec_kernel")[0]))
        _lib.lib().tfc_profile_enable(0)
        assert torch.equal(dec.reshape(S, E), vt) and bool(ok.all())
    enc, dec_ms = min(r[0] for r in res), min(r[1] for r in res)
    wide_share = float((width[it.cpu().numpy()] > 64).mean())
    print(f"{kind:7s} wide rows {wide_share:6.3f}  bits/sym {blob.numel() * 8 / (S * E):5.2f}  "
          f"encode {enc:7.3f} ms = {enc * 2.4e6 / E:6.1f} cyc/sym   decode {dec_ms:7.3f} ms = {dec_ms * 2.4e6 / E:6.1f} cyc/sym")
