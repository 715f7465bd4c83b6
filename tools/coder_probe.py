#!/usr/bin/env python
"""Kernel-time probe for the range coder: cycles/symbol for table sets of different widths.
   python tools/coder_probe.py            (on a GPU box)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compression_amd as tfc  # noqa: E402
from compression_amd import _lib, synthetic  # noqa: E402


def q(name):
    ms, n = C.c_double(), C.c_int64()
    _lib.lib().tfc_profile_query(name.encode(), C.byref(ms), C.byref(n))
    return ms.value / max(n.value, 1)


def run(label, sigma0, octave, streams=512, elems=49152, ntab=192, esc=0.0, reps=3):
    pmfs, _ = synthetic.gaussian_pmfs(num_tables=ntab, sigma0=sigma0, octave=octave)
    dev = torch.device("cuda")
    cdfs = [tfc.pmf_to_quantized_cdf(torch.from_numpy(p).to(dev), 12).cpu().numpy() for p in pmfs]
    lookup = synthetic.assemble_lookup(cdfs, 12, overflow=True)
    value = synthetic.sample_symbols(lookup, streams, elems, seed=0, escape_fraction=esc)
    lt, vt = torch.from_numpy(lookup), torch.from_numpy(value).to(dev)
    widths = [len(c) - 1 for c in cdfs]
    for r in range(reps + 1):
        if r == 1:
            _lib.lib().tfc_profile_enable(1)
        h = tfc.create_range_encoder([streams], lt)
        h = tfc.entropy_encode_channel(h, vt)
        blob, offs = tfc.gen_ops._finalize_device(h)
        d = tfc.create_range_decoder((blob, offs, (streams,)), lt)
        d, out = tfc.entropy_decode_channel(d, [elems], torch.int32)
        ok = tfc.entropy_decode_finalize(d)
    torch.cuda.synchronize()
    e, dd = q("enc_kernel"), q("dec_kernel")
    _lib.lib().tfc_profile_enable(0)
    assert torch.equal(out, vt) and bool(ok.all())
    ghz = 2.4
    print(f"{label:34s} widths {min(widths):4d}..{max(widths):4d}  enc {e:7.3f} ms ({e*1e-3*ghz*1e9/elems:6.1f} cyc/sym @2.4GHz)"
          f"  dec {dd:7.3f} ms ({dd*1e-3*ghz*1e9/elems:6.1f} cyc/sym)  bits/sym {8*int(offs[-1])/value.size:.2f}")


if __name__ == "__main__":
    run("C2 (sigma .25..63)", 0.25, 24.0)
    run("narrow only (sigma .25..8)", 0.25, 38.0)
    run("wide only (sigma 20..63)", 20.0, 115.0)
    run("mid (sigma 11..21, 65-128 syms)", 11.0, 200.0)
    run("C2 + 1% escapes", 0.25, 24.0, esc=0.01)
    run("narrow, 4096 streams", 0.25, 38.0, streams=4096, elems=12288)
