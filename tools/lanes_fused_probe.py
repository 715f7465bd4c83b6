#!/usr/bin/env python
"""Lane kernels on bottleneck values against plain int32 symbols (BASELINE config 2 geometry, 20 batches per
launch): tfc_encoder_encode_quantized_many / tfc_decoder_decode_dequantized_many (elementwise quantise pass +
the int32 blocks) must cost what int32 channel mode costs.  Prints milliseconds per 20-batch launch group."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import compression_amd as tfc
from compression_amd import _lib, synthetic

dev = torch.device("cuda", 0)
lookup = bench.build_tables(dev)
lt = torch.from_numpy(lookup)
rows = synthetic.lookup_rows(lookup)
N, S, E = 20, bench.STREAMS, bench.ELEMS
syms = [bench.sample_symbols_device(lookup, k, dev) for k in range(N)]
off = torch.tensor([-(len(c) - 2) // 2 for _, c in rows], dtype=torch.int32, device=dev)       # any offsets: y = sym + offset
for dtype in (torch.float32, torch.bfloat16):
    ys = [(s + off.repeat(E // len(rows))).to(dtype) for s in syms]
    code = {torch.float32: 0, torch.bfloat16: 1}[dtype]

    def run(fused):
        hs = tfc.create_range_encoders(N, [S], lt, mode="throughput", deferred_errors=True)
        if fused:
            hp = (C.c_void_p * N)(*[h.ptr for h in hs])
            yp = (C.c_void_p * N)(*[y.data_ptr() for y in ys])
            _lib.check(_lib.lib().tfc_encoder_encode_quantized_many(N, hp, yp, code, None, off.data_ptr(), len(rows), E,
                                                                    _lib.stream_ptr()))
        else:
            hs = tfc.entropy_encode_channel_many(hs, syms)
        hs = tfc.entropy_encode_finalize_device_many(hs)
        ds = tfc.create_range_decoders(hs, lt, mode="throughput")
        if fused:
            outs = [torch.empty(S, E, dtype=dtype, device=dev) for _ in range(N)]
            dp = (C.c_void_p * N)(*[d.ptr for d in ds])
            op = (C.c_void_p * N)(*[o.data_ptr() for o in outs])
            _lib.check(_lib.lib().tfc_decoder_decode_dequantized_many(N, dp, op, code, None, off.data_ptr(), len(rows), E,
                                                                      _lib.stream_ptr()))
        else:
            ds, outs = tfc.entropy_decode_channel_many(ds, [E])
        ok = tfc.entropy_decode_finalize_device_many(ds)
        return hs, outs, ok

    for fused in (False, True):
        run(fused)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hs, outs, ok = run(fused)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        good = all(torch.equal(o.reshape(S, E), (ys[k] if fused else syms[k])) for k, o in enumerate(outs)) and bool(ok.all())
        print(f"{str(dtype):16s} {'fused values' if fused else 'int32 symbols'}: {1e3 * dt:7.2f} ms per {N}-batch group, round trip exact: {good}")
