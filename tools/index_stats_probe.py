#!/usr/bin/env python
"""On the GPU box: how the scale indexes of the bmshj2018 main latent (the bench's model and images) fall on
narrow (<= 64 symbols) and wide tables — per symbol, per 8-symbol block and per 64-symbol batch of a stream
in coding order — i.e. how often the wave-per-stream decoder takes its two-stage step and how much a finer
decision could save.  Usage: python tools/index_stats_probe.py [batch]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compression_amd as tfc
from compression_amd import synthetic

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
model = tfc.models.BMSHJ2018Model(num_filters=192, compute_dtype=torch.bfloat16).cuda().init_compression()
x = torch.from_numpy(synthetic.lowpass_images(batch, 512, 768)).cuda()
with torch.no_grad():
    xf = x.to(torch.bfloat16)
    y = model.analysis_transform(xf)
    z = model.hyper_analysis_transform(torch.abs(y))
    z_hat = model.side_entropy_model.quantize(z)
    idx = model.hyper_synthesis_transform(z_hat)[:, :y.shape[1], :y.shape[2], :]
em = model.entropy_model
flat = em._flatten_indexes(em._normalize_indexes(idx.float())).reshape(batch, -1).cpu().numpy()
cdf = em.cdf.numpy()
# ragged rows: [precision, 0, ..., 1 << p]; row starts from the directory of lengths
widths, pos = [], 0
while pos < cdf.size:
    prec = abs(int(cdf[pos]))
    end = pos + 2
    while cdf[end] != (1 << prec) or (end + 1 < cdf.size and cdf[end + 1] == (1 << prec)):
        end += 1
    widths.append(end - pos - 1)          # symbols of the row
    pos = end + 1
widths = np.array(widths)
print("tables:", len(widths), "widths min/median/max", widths.min(), int(np.median(widths)), widths.max(),
      "wide (> 64 symbols):", int((widths > 64).sum()))
hist = np.bincount(flat.reshape(-1), minlength=len(widths))
print("index histogram (per cent):", np.round(100 * hist / hist.sum(), 1).tolist())
wide = widths[flat] > 64
n = wide.shape[1] // 64 * 64
w = wide[:, :n]
print(f"symbols on wide tables: {100 * w.mean():.1f} %")
print(f"8-symbol blocks with a wide table: {100 * w.reshape(batch, -1, 8).any(-1).mean():.1f} %")
print(f"64-symbol batches with a wide table: {100 * w.reshape(batch, -1, 64).any(-1).mean():.1f} %")
