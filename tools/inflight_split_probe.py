#!/usr/bin/env python
"""Encode-only and decode-only steps in flight (D host threads x HIP streams), against the VALU-issue
floor of each kernel — separates "the two kernels disturb each other" (e.g. the shared 64 KB instruction
cache of a CU pair: decoder 56 KB + encoder 22 KB of code) from "one kernel does not fill the SIMDs".
python tools/inflight_split_probe.py  (on a GPU box; GPU_MAX_HW_QUEUES=16 as in bench.py)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from concurrent.futures import ThreadPoolExecutor
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import compression_amd as tfc
from compression_amd import synthetic

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
lookup = bench.build_tables(dev)
value = synthetic.sample_symbols(lookup, bench.STREAMS, bench.ELEMS, seed=0)
lt, vt = torch.from_numpy(lookup), torch.from_numpy(value).to(dev)
blob0, off0, dec0, ok0 = bench.one_step(lt, vt)
torch.cuda.synchronize()


def enc_only():
    h = tfc.create_range_encoder([bench.STREAMS], lt)
    h = tfc.entropy_encode_channel(h, vt)
    return tfc.gen_ops._finalize_device(h)


def dec_only():
    d = tfc.create_range_decoder((blob0, off0, (bench.STREAMS,)), lt)
    d, decoded = tfc.entropy_decode_channel(d, [bench.ELEMS], torch.int32)
    return tfc.entropy_decode_finalize(d)


def worker(stream, fn, n):
    torch.cuda.set_device(0)
    with torch.cuda.stream(stream):
        for _ in range(n):
            fn()
        stream.synchronize()


FLOOR = {"encode": 0.611, "decode": 1.183}   # ms: SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / 2.4 GHz
for name, fn in (("encode", enc_only), ("decode", dec_only)):
    for D in (1, 2, 4, 6, 8, 12):
        K = 4 * D
        streams = [torch.cuda.Stream() for _ in range(D)]
        with ThreadPoolExecutor(D) as pool:
            list(pool.map(lambda s: worker(s, fn, 1), streams))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            list(pool.map(lambda s: worker(s, fn, K // D), streams))
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f"{name} only, in flight {D:2d}: {dt * 1e3 / K:6.3f} ms/step   VALU floor {FLOOR[name]:.3f} ms "
              f"-> {FLOOR[name] / (dt * 1e3 / K):.2f} of it")
