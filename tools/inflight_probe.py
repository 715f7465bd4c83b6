#!/usr/bin/env python
"""Throughput of the C2 round trip with D independent steps in flight (one host thread and
one HIP stream each).  python tools/inflight_probe.py  (on a GPU box)"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from compression_amd import synthetic

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
lookup = bench.build_tables(dev)
value = synthetic.sample_symbols(lookup, bench.STREAMS, bench.ELEMS, seed=0)
lt, vt = torch.from_numpy(lookup), torch.from_numpy(value).to(dev)
for _ in range(2):
    bench.one_step(lt, vt)
torch.cuda.synchronize()

def worker(stream, n):
    torch.cuda.set_device(0)
    with torch.cuda.stream(stream):
        for _ in range(n):
            blob, off, dec, ok = bench.one_step(lt, vt)
        stream.synchronize()
    return dec, ok

for D in (1, 2, 3, 4, 6, 8):
    K = 4 * D
    streams = [torch.cuda.Stream() for _ in range(D)]
    with ThreadPoolExecutor(D) as pool:
        list(pool.map(lambda s: worker(s, 1), streams))      # warm each stream
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = list(pool.map(lambda s: worker(s, K // D), streams))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    okall = all(bool(ok.all()) and torch.equal(dec.reshape(bench.STREAMS, bench.ELEMS), vt) for dec, ok in res)
    mp = K * bench.STREAMS * bench.PIXELS_PER_STREAM / 1e6 / dt
    print(f"in flight {D}: {K} steps in {dt*1e3:7.1f} ms  -> {dt*1e3/K:6.2f} ms/step, {mp:8.0f} Mpixels/s, exact={okall}")
