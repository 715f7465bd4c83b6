#!/usr/bin/env python
"""Where a conv3_bf16_kernel workgroup's time goes and how the workgroups of a launch line up in time: a library built
with -DTFC_CONV3_EXP=64 (tools/conv3_variants.sh build 64) records per workgroup the 100 MHz clock at entry, behind the
prologue's barrier, at the end of the K loop, with the stores issued, with them acknowledged, and the CU.
Usage (GPU box): TFC_LIB_PATH=tools/probe_libs/libtfc_conv3_exp64.so python tools/conv3_clock_probe.py [batch] [down|up]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compression_amd._lib as _L
_L.LIB_PATH = os.environ.get("TFC_LIB_PATH", os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe_libs",
                                                          "libtfc_conv3_exp64.so"))
from compression_amd.layers import conv2d_down, conv2d_up
from compression_amd.layers.functional import GDNPrepared, conv2d_gdn

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
which = sys.argv[2] if len(sys.argv) > 2 else "down"
dev = "cuda"
gen = torch.Generator().manual_seed(3)
gdn = which.endswith("_gdn")            # down_gdn / up_gdn: GDN / IGDN as the layer's activation, inside the kernel
which = which.split("_")[0]
if which == "down":
    fn, (H, W) = conv2d_down, (256, 384)
else:
    fn, (H, W) = conv2d_up, (128, 192)
if gdn:
    prepared = GDNPrepared(torch.rand(192) + 1.0, torch.rand(192, 192) * 0.01 + 0.1 * torch.eye(192), torch.bfloat16)
    def fn(x, w, bias, stride, act, up=(which == "up")):
        y, fused = conv2d_gdn(x, w, bias, stride, up, prepared, up)
        assert fused
        return y
x = torch.randn(batch, H, W, 192, generator=gen).to(torch.bfloat16).to(dev)
w = (torch.randn(5, 5, 192, 192, generator=gen) / (25 * 192) ** 0.5).to(dev)
bias = torch.randn(192, generator=gen).to(dev)
os.environ["TFC_CONV_GEN"] = "4"
for _ in range(3):
    y = fn(x, w, bias, 2, None)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
y = fn(x, w, bias, 2, None)
b.record()
torch.cuda.synchronize()
print(f"{which} layer, {batch} images: {a.elapsed_time(b):.3f} ms (the launch the clocks below are of; the up layer: its LAST launch)")
lib = C.CDLL(_L.LIB_PATH)
blocks = batch * ((W // (2 if which == "down" else 1) + 31) // 32) * ((H // (2 if which == "down" else 1) + 7) // 8)
wgs = min(blocks, 16384)
buf = (C.c_ulonglong * (8 * wgs))()
assert lib.tfc_debug_conv3_clocks(buf, wgs) == wgs
t = np.frombuffer(buf, dtype=np.uint64).reshape(wgs, 8).astype(np.int64)
t = t[t[:, 0] > 0]
wgs = len(t)
t0 = t[:, 0].min()
us = (t[:, :5] - t0) / 100.0
hw = t[:, 5]
xcc, se, cu = (hw >> 32) & 0xF, (hw >> 13) & 0x7, (hw >> 8) & 0xF
cuid = xcc * 64 + se * 16 + cu
pro, kloop, issue, drain = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2], us[:, 4] - us[:, 3]
q = lambda v: "%.1f / %.1f / %.1f / %.1f" % tuple(np.percentile(v, [10, 50, 90, 99]))
print(f"{wgs} workgroups on {len(np.unique(cuid))} CUs, {us[:, 4].max():.1f} us from the first entry to the last acknowledged store")
print("per workgroup, us (10 / 50 / 90 / 99 %):")
print("  prologue (entry -> first patch + weights in LDS)   ", q(pro))
print("  K loop                                              ", q(kloop))
print("  epilogue: stores issued                             ", q(issue))
if gdn:
    print("  of the epilogue, GDN stage: y packed (+ image copied) ", q((t[:, 6] - t[:, 2]) / 100.0))
    print("                              contraction               ", q((t[:, 7] - t[:, 6]) / 100.0))
    print("                              y / norm + stores issued  ", q((t[:, 3] - t[:, 7]) / 100.0))
print("  stores acknowledged                                 ", q(drain))
print("  whole workgroup                                     ", q(us[:, 4] - us[:, 0]))
# the gap on a CU between a workgroup's end and the next one's entry
gaps = []
for c in np.unique(cuid):
    m = np.where(cuid == c)[0]
    o = m[np.argsort(us[m, 0])]
    gaps.extend(us[o[1:], 0] - us[o[:-1], 4])
if gaps:
    print("  CU idle between a workgroup's last clock and the next one's entry", q(np.array(gaps)))
# how many workgroups are in their epilogue at the same moment
grid = np.linspace(0, us[:, 4].max(), 2000)
inepi = ((us[:, 2][None, :] <= grid[:, None]) & (grid[:, None] < us[:, 4][None, :])).sum(1)
live = ((us[:, 0][None, :] <= grid[:, None]) & (grid[:, None] < us[:, 4][None, :])).sum(1)
print("workgroups in their epilogue at a moment: mean %.1f, 90 %% %.0f, max %d of mean %.0f resident" % (
    inepi.mean(), np.percentile(inepi, 90), inepi.max(), live.mean()))
starts = np.sort(us[:, 0])
print("entries in the first 400 us, 20-us bins:", np.histogram(starts, bins=np.arange(0, 420, 20))[0].tolist())

# where the first wave waited inside the K loop (core-clock cycles, tfc_debug_conv3_waits)
ksteps = 12 * (25 if which == "down" else 4)      # K steps of an item (the up layer: of its last launch, the 4-tap phase)
wb = (C.c_ulonglong * (8 * wgs))()
if hasattr(lib, "tfc_debug_conv3_waits") and lib.tfc_debug_conv3_waits(wb, wgs) == wgs:
    wv = np.frombuffer(wb, dtype=np.uint64).reshape(wgs, 8).astype(np.float64)
    wv = wv[wv[:, 6] > 0]
    tot = np.median(wv[:, 6])
    print("K loop of the first wave: %.0f core-clock cycles (median) = %.2f GHz against the 100 MHz clock; of them waiting" % (
        tot, tot / (np.median(kloop) * 1e3)))
    # per instruction: weight chunks of STAGE pieces (8 at 192 channels x 5 taps), patches of NPG pieces
    per_step = np.median(wv[:, 7]) / max(1, ksteps)
    print("   a pair of clock reads by itself: %.0f cycles" % per_step)
    for i, name in enumerate(("weight requests (buffer_load_dwordx4, 1 KB contiguous)", "patch requests (buffer_load_dwordx4, 64 lanes x 16 B gathered)",
                              "weight ds_write_b128", "patch ds_write_b128")):
        print("   issuing the %s: %5.1f %% of the K loop" % (name, 100 * np.median(wv[:, i]) / tot))
    print("   waiting at a chunk's start for its weights: %5.1f %%" % (100 * np.median(wv[:, 4]) / tot))
    print("   barriers: %5.1f %%" % (100 * np.median(wv[:, 5]) / tot))
    print("   raw medians (cycles):", [int(np.median(wv[:, i])) for i in range(8)])
