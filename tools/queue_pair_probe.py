#!/usr/bin/env python
"""Which pairs of HIP streams run a SMALL-grid long kernel (the wave-per-stream decoder: 128 workgroups, ~4 ms)
concurrently with a chain of LARGE-grid kernels (elementwise over 1 GiB)?  Hypothesis from the C4 pipeline's
timeline: a queue's small kernel is not dispatched while another queue ON THE SAME HARDWARE PIPE is still
launching the workgroups of a large grid.  Creates several (coder-masked, transform-masked) stream pairs plus
ordinary streams and prints alone / together times per pair."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", sys.argv[1] if len(sys.argv) > 1 else "16")
import bench
import compression_amd as tfc
from compression_amd import pipeline

dev = torch.device("cuda", 0)
lookup = bench.build_tables(dev)
lt = torch.from_numpy(lookup)
sym = bench.sample_symbols_device(lookup, 0, dev)
h = tfc.create_range_encoder([bench.STREAMS], lt, mode="latency")
h = tfc.entropy_encode_channel(h, sym)
blob, off = tfc.gen_ops._finalize_device(h)
big = torch.zeros(1 << 28, device=dev)


def coder_work():
    d = tfc.create_range_decoder((blob, off, (bench.STREAMS,)), lt, mode="latency")
    d, dec = tfc.entropy_decode_channel(d, [bench.ELEMS], torch.int32)
    return d, dec


def big_work():
    for _ in range(14):
        big.add_(1.0)


def timed(pairs):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    keep = []
    for s, f in pairs:
        with torch.cuda.stream(s):
            keep.append(f())
    for s, _ in pairs:
        s.synchronize()
    return 1e3 * (time.perf_counter() - t0)


from compression_amd.layers import conv2d_down, gdn_forward
xc = torch.randn(128, 192, 128, 192, device=dev).bfloat16()
wc = (torch.randn(5, 5, 192, 192) / 70).to(dev)
bc = torch.zeros(192, device=dev)
xg = torch.randn(128 * 192 * 128, 192, device=dev).bfloat16()
beta = torch.ones(192, device=dev)
gamma = (0.1 * torch.eye(192)).to(dev)


def conv_work():
    return [conv2d_down(xc, wc, bc, 2) for _ in range(3)]


def gdn_work():
    return [gdn_forward(xg, beta, gamma) for _ in range(4)]


cands = []
for cus in (128, 32):
    p = pipeline.CoderPartition(coder_cus=cus, depth=1)
    cands.append((f"masked{cus}", p.lane(0).transform, p.lane(0).coder))
cands.append(("plain", torch.cuda.Stream(), torch.cuda.Stream()))
cands.append(("plain-hipri", torch.cuda.Stream(), torch.cuda.Stream(priority=-1)))
p32 = pipeline.CoderPartition(coder_cus=32, depth=1)
cands.append(("tmask-hipri", p32.lane(0).transform, torch.cuda.Stream(priority=-1)))
cands.append(("tmask-plain", p32.lane(0).transform, torch.cuda.Stream()))
for name, t, c in cands:
    for wname, work in (("elementwise", big_work), ("conv", conv_work), ("gdn", gdn_work)):
        for s, f in ((t, work), (c, coder_work)):
            timed([(s, f)])
        a = timed([(t, work)])
        b = timed([(c, coder_work)])
        both = timed([(t, work), (c, coder_work)])
        both2 = timed([(c, coder_work), (t, work)])
        print(f"{name:9s} {wname:11s} {a:6.2f}  coder {b:6.2f}  together {both:6.2f} / {both2:6.2f}  (max {max(a, b):.2f}, sum {a + b:.2f})")


# T-first order, the transform chain on one stream and the coder on each of many other streams: does ANY pair overlap?
print("pair scan (transform work enqueued first): conv on stream 0, coder on stream k")
streams = [torch.cuda.Stream() for _ in range(int(os.environ.get("TFC_SCAN_STREAMS", "14")))]
for s_ in streams:
    timed([(s_, coder_work)])
timed([(streams[0], conv_work)])
a = timed([(streams[0], conv_work)])
b = timed([(streams[1], coder_work)])
res = []
for k in range(1, len(streams)):
    res.append(timed([(streams[0], conv_work), (streams[k], coder_work)]))
print(f"conv alone {a:.2f}, coder alone {b:.2f}, together by k: " + " ".join(f"{r:.2f}" for r in res))


# Gating: the coder kernel waits for E0 (an earlier burst on a third stream) and records a gate right before it is
# launched; the transform burst waits for that gate.  Do the two then overlap (coder first), whatever the host order?
print("gated release: X = earlier burst; coder waits X; conv burst waits the gate recorded in front of the coder kernel")
p128 = pipeline.CoderPartition(coder_cus=128, depth=1)
T, Cs, X = p128.lane(0).transform, p128.lane(0).coder, torch.cuda.Stream()
for variant in ("gated", "gated+dummy", "ungated"):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(X):
            big_work()
            e0 = torch.cuda.Event(); e0.record(X)
        g = torch.cuda.Event()
        with torch.cuda.stream(Cs):
            Cs.wait_event(e0)
            g.record(Cs)
            keep = coder_work()
        with torch.cuda.stream(T):
            if variant != "ungated":
                T.wait_event(g)
            else:
                T.wait_event(e0)
            if variant == "gated+dummy":
                small2 = torch.zeros(64, device=dev) + 1
            keep2 = conv_work()
        torch.cuda.synchronize()
        dt = 1e3 * (time.perf_counter() - t0)
    print(f"{variant:12s} total {dt:6.2f} ms")
a = timed([(X, big_work)]); b = timed([(Cs, coder_work)]); c = timed([(T, conv_work)])
print(f"alone: X {a:.2f}, coder {b:.2f}, conv {c:.2f}: ideal {a + max(b, c):.2f}, serial {a + b + c:.2f}")
