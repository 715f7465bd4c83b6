"""Tiny lane-kernel smoke: encode/decode a few streams in throughput mode against the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import compression_amd as tfc
from compression_amd import synthetic
from oracle import oracle
port = oracle.port()
pmfs, _ = synthetic.gaussian_pmfs(num_tables=16, octave=2.0)
cdfs = [port.pmf_to_quantized_cdf(p, 12) for p in pmfs]
lookup = synthetic.assemble_lookup(cdfs, 12, overflow=True)
lt = torch.from_numpy(lookup)
for frac in (0.0, 0.05):
    value = synthetic.sample_symbols(lookup, 8, 2048, seed=0, escape_fraction=frac)
    want, _, _ = port.encode(lookup, value, threads=2)
    print("encode", frac, flush=True)
    h = tfc.create_range_encoder([8], lt, mode="throughput")
    h = tfc.entropy_encode_channel(h, torch.from_numpy(value).cuda())
    torch.cuda.synchronize()
    print("finalize", flush=True)
    got = tfc.entropy_encode_finalize(h)
    same = [bytes(s) for s in got] == want
    print("bytes identical:", same, [len(s) for s in got][:4], [len(s) for s in want][:4], flush=True)
    hd = tfc.create_range_decoder(np.array(want, dtype=object), lt, mode="throughput")
    hd, dec = tfc.entropy_decode_channel(hd, [2048], torch.int32)
    torch.cuda.synchronize()
    print("decoded", flush=True)
    ok = tfc.entropy_decode_finalize(hd)
    bad = np.argwhere(dec.cpu().numpy() != value)
    print("decode exact:", len(bad) == 0, bad[:5].tolist(), "ok flags", bool(ok.all()), flush=True)
