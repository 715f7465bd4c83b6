#!/usr/bin/env python
"""Index-mode round trip (ContinuousIndexedEntropyModel — every symbol carries its own
table index, as in bmshj2018's conditional Gaussian) on the bench geometry: 512 streams x 49152 symbols,
64 scale tables, indexes drawn per element.  Serial and with steps in flight (throughput mode), bit-exact
decode checked.  python tools/index_mode_probe.py  (on a GPU box)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from concurrent.futures import ThreadPoolExecutor
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import compression_amd as tfc
from compression_amd import synthetic

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
pmfs, minima = synthetic.gaussian_pmfs(num_tables=64, octave=8.0)     # sigma 0.25 .. ~59, 64 tables
cdfs = [tfc.pmf_to_quantized_cdf(torch.from_numpy(p).to(dev), 12).cpu().numpy() for p in pmfs]
lookup = synthetic.assemble_lookup(cdfs, 12, overflow=True)
rows = synthetic.lookup_rows(lookup)
rng = np.random.default_rng(0)
S, E = bench.STREAMS, bench.ELEMS
index = rng.integers(0, len(rows), (S, E)).astype(np.int32)
value = np.zeros((S, E), np.int32)
for t, (sp, cdf) in enumerate(rows):
    m = index == t
    u = rng.integers(0, 1 << 12, int(m.sum()))
    value[m] = np.minimum(np.searchsorted(cdf, u, side="right") - 1, len(cdf) - 3)
lt = torch.from_numpy(lookup)
vt, it = torch.from_numpy(value).to(dev), torch.from_numpy(index).to(dev)


def step():
    h = tfc.create_range_encoder([S], lt)
    h = tfc.entropy_encode_index(h, it, vt)
    blob, off = tfc.gen_ops._finalize_device(h)
    d = tfc.create_range_decoder((blob, off, (S,)), lt)
    d, dec = tfc.entropy_decode_index(d, it, [E], torch.int32)
    ok = tfc.entropy_decode_finalize(d)
    return blob, dec, ok


blob, dec, ok = step()
assert torch.equal(dec.reshape(S, E), vt) and bool(ok.all())
print(f"{len(rows)} tables, {blob.numel() * 8 / (S * E):.3f} bits/symbol")


def worker(stream, n):
    torch.cuda.set_device(0)
    with torch.cuda.stream(stream):
        for _ in range(n):
            step()
        stream.synchronize()


for D in (1, 8, 12):
    tfc.set_default_mode("throughput" if D > 1 else "latency")
    K = 4 * D
    streams = [torch.cuda.Stream() for _ in range(D)]
    with ThreadPoolExecutor(D) as pool:
        list(pool.map(lambda s: worker(s, 1), streams))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        list(pool.map(lambda s: worker(s, K // D), streams))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"index mode, in flight {D:2d}: {dt * 1e3 / K:6.3f} ms/step = {S * E / 1e9 / (dt / K):6.2f} Gsymbols/s "
          f"({S * bench.PIXELS_PER_STREAM / 1e6 / (dt / K):8.0f} Mpixels/s at 0.75 symbols per pixel)")
tfc.set_default_mode("auto")
