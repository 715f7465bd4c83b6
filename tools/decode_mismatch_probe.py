import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import compression_amd as tfc
from compression_amd import synthetic

def trial(label, lookup, value):
    lt, vt = torch.from_numpy(lookup), torch.from_numpy(value).cuda()
    streams = value.shape[0]
    os.environ["TFC_FORCE_GENERIC"] = "1"
    h = tfc.create_range_encoder([streams], lt); h = tfc.entropy_encode_channel(h, vt)
    enc = tfc.entropy_encode_finalize(h)
    os.environ["TFC_FORCE_GENERIC"] = "0"
    d = tfc.create_range_decoder(enc, lt)
    d, out = tfc.entropy_decode_channel(d, [value.shape[1]], torch.int32)
    ok = tfc.entropy_decode_finalize(d)
    out = out.cpu().numpy()
    bad = np.argwhere(out != value)
    first = tuple(bad[0]) if len(bad) else None
    print(f"{label:40s} mismatches {len(bad):7d} first {first} finalize_ok {bool(ok.all())}")
    if first is not None:
        s, j = first
        print("   want", value[s, max(0, j - 4):j + 8].tolist()); print("   got ", out[s, max(0, j - 4):j + 8].tolist())

pm, _ = synthetic.gaussian_pmfs(num_tables=4, sigma0=0.5, octave=2.0)
cd = [tfc.pmf_to_quantized_cdf(torch.from_numpy(p).cuda(), 12).cpu().numpy() for p in pm]
look = synthetic.assemble_lookup(cd, 12, overflow=False)
for n in (16, 40, 64, 100, 128, 640):
    v = synthetic.sample_symbols(look, 3, n, seed=n)
    trial(f"narrow 4 tables, {n} symbols", look, v)
pm, _ = synthetic.gaussian_pmfs(num_tables=8, sigma0=10.0, octave=4.0)
cd = [tfc.pmf_to_quantized_cdf(torch.from_numpy(p).cuda(), 12).cpu().numpy() for p in pm]
look = synthetic.assemble_lookup(cd, 12, overflow=True)
for n in (40, 64, 256):
    v = synthetic.sample_symbols(look, 3, n, seed=n)
    trial(f"wide 8 tables, {n} symbols", look, v)
    v = synthetic.sample_symbols(look, 3, n, seed=n, escape_fraction=0.05)
    trial(f"wide 8 tables + escapes, {n} symbols", look, v)
