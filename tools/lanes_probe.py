"""Times the C2 round trip (512 streams x 49152 symbols) in both kernel families: one call alone, and
B independent steps in flight on B HIP streams driven by ONE host thread through the stream-ordered
path (deferred errors, device finalize, decoder on the encoder's device-resident strings).
Usage: python tools/lanes_probe.py [--inflight 1,4,8,16,32] [--escape-fraction 0.0]"""
import argparse
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

import compression_amd as tfc
from compression_amd import synthetic


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--inflight", default="1,2,4,8,16,32")
    ap.add_argument("--escape-fraction", type=float, default=0.0)
    ap.add_argument("--streams", type=int, default=512)
    ap.add_argument("--elems", type=int, default=16 * 16 * 192)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--modes", default="latency,throughput")
    args = ap.parse_args()

    pmfs, _ = synthetic.gaussian_pmfs(192)
    cdfs = [tfc.pmf_to_quantized_cdf(torch.from_numpy(p), 12).cpu().numpy() for p in pmfs]
    lookup = synthetic.assemble_lookup(cdfs, 12, overflow=True)
    lt = torch.from_numpy(lookup)
    S, E = args.streams, args.elems
    depth_max = max(int(x) for x in args.inflight.split(","))
    nvals = min(depth_max, 8)
    vals = [torch.from_numpy(synthetic.sample_symbols(lookup, S, E, seed=k, escape_fraction=args.escape_fraction)).cuda()
            for k in range(nvals)]
    torch.cuda.synchronize()
    pixels = S * 256 * 256 if E == 49152 else S * E

    for mode in args.modes.split(","):
        # one call alone, kernel times by events
        for rep in range(2):
            e0, e1, e2, e3 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
            h = tfc.create_range_encoder([S], lt, mode=mode, deferred_errors=True)
            e0.record()
            h = tfc.entropy_encode_channel(h, vals[0])
            e1.record()
            h = tfc.entropy_encode_finalize_device(h)
            d = tfc.create_range_decoder(h, lt, mode=mode)
            e2.record()
            d, out = tfc.entropy_decode_channel(d, [E], torch.int32)
            e3.record()
            ok = tfc.entropy_decode_finalize_device(d)
            torch.cuda.synchronize()
            total = tfc.entropy_encode_status(h)
            assert bool((out == vals[0]).all()) and bool(ok.all())
        print(f"[{mode}] alone: encode {e0.elapsed_time(e1):.3f} ms, finalize+open {e1.elapsed_time(e2):.3f} ms, "
              f"decode {e2.elapsed_time(e3):.3f} ms, {total} bytes ({8 * total / (S * E):.3f} bits/symbol)", flush=True)

        for depth in [int(x) for x in args.inflight.split(",")]:
            streams = [torch.cuda.Stream() for _ in range(depth)]
            best = None
            for rep in range(args.reps):
                keep = []
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                steps = max(depth, 2 * depth if depth <= 8 else depth)
                for k in range(steps):
                    st = streams[k % depth]
                    with torch.cuda.stream(st):
                        v = vals[k % nvals]
                        h = tfc.create_range_encoder([S], lt, mode=mode, deferred_errors=True)
                        h = tfc.entropy_encode_channel(h, v)
                        h = tfc.entropy_encode_finalize_device(h)
                        d = tfc.create_range_decoder(h, lt, mode=mode)
                        d, out = tfc.entropy_decode_channel(d, [E], torch.int32)
                        ok = tfc.entropy_decode_finalize_device(d)
                        keep.append((h, d, out, ok, k % nvals))
                t_host = time.perf_counter() - t0
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                for h, d, out, ok, kv in keep[-2:]:
                    tfc.entropy_encode_status(h)
                    assert bool((out == vals[kv]).all()) and bool(ok.all())
                del keep
                best = dt / steps if best is None else min(best, dt / steps)
            print(f"[{mode}] {depth:3d} in flight: {1e3 * best:.3f} ms/step = {pixels / best / 1e6:.0f} Mpixels/s "
                  f"(host enqueue {1e3 * t_host / steps:.3f} ms/step)", flush=True)


if __name__ == "__main__":
    main()
