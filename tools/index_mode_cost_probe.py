#!/usr/bin/env python
"""Index mode against channel mode on the pipelined kernels (BASELINE config 2 geometry with the 64 tables of a
hyperprior model — the 192 tables of config 2 leave no LDS for the decoder's index window — 20 batches per launch): the
same symbols coded with the table of every element taken from an index tensor (here the element's channel, so that the
bytes must equal channel mode's) — EntropyEncodeIndex / EntropyDecodeIndex with quantise / dequantise fused, the call
the hyperprior models make (continuous_indexed.py:355-417) — and as int32 channel-mode symbols.  Prints milliseconds
per 20-batch launch group and their ratio."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import bench
import compression_amd as tfc
from compression_amd import _lib, synthetic

dev = torch.device("cuda", 0)
lookup = bench.build_tables(dev, 64, 8.0)
lt = torch.from_numpy(lookup)
rows = synthetic.lookup_rows(lookup)
N, S, E = 20, bench.STREAMS, bench.ELEMS
syms = [bench.sample_symbols_device(lookup, k, dev) for k in range(N)]
zero = torch.zeros(len(rows), dtype=torch.int32, device=dev)
index = (torch.arange(E, device=dev, dtype=torch.int32) % len(rows)).repeat(S, 1).contiguous()
ys = [s.to(torch.float32) for s in syms]


def run(indexed):
    hs = tfc.create_range_encoders(N, [S], lt, mode="throughput", deferred_errors=True)
    if indexed:
        hp = (C.c_void_p * N)(*[h.ptr for h in hs])
        yp = (C.c_void_p * N)(*[y.data_ptr() for y in ys])
        ip = (C.c_void_p * N)(*[index.data_ptr()] * N)
        _lib.check(_lib.lib().tfc_encoder_encode_quantized_indexed_many(N, hp, yp, 0, ip, zero.data_ptr(), E, _lib.stream_ptr()))
    else:
        hs = tfc.entropy_encode_channel_many(hs, syms)
    hs = tfc.entropy_encode_finalize_device_many(hs)
    ds = tfc.create_range_decoders(hs, lt, mode="throughput")
    if indexed:
        outs = [torch.empty(S, E, dtype=torch.float32, device=dev) for _ in range(N)]
        dp = (C.c_void_p * N)(*[d.ptr for d in ds])
        op = (C.c_void_p * N)(*[o.data_ptr() for o in outs])
        ip = (C.c_void_p * N)(*[index.data_ptr()] * N)
        _lib.check(_lib.lib().tfc_decoder_decode_dequantized_indexed_many(N, dp, ip, op, 0, zero.data_ptr(), E, _lib.stream_ptr()))
    else:
        ds, outs = tfc.entropy_decode_channel_many(ds, [E])
    ok = tfc.entropy_decode_finalize_device_many(ds)
    return hs, outs, ok


times, blobs = {}, {}
for indexed in (False, True):
    run(indexed)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        hs, outs, ok = run(indexed)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    good = all(torch.equal(o.reshape(S, E).to(torch.int32), syms[k]) for k, o in enumerate(outs)) and bool(ok.all())
    blobs[indexed] = [tfc.device_strings(h)[0].clone() for h in hs[:2]]
    times[indexed] = best
    print(f"{'index mode, float32 values (quantise / dequantise fused)' if indexed else 'channel mode, int32 symbols':58s}: "
          f"{1e3 * best:7.2f} ms per {N}-batch group, round trip exact: {good}", flush=True)
same = all(torch.equal(a, b) for a, b in zip(blobs[False], blobs[True]))
print(f"index / channel = {times[True] / times[False]:.3f}; bytes identical: {same}")
