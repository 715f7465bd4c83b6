import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import compression_amd as tfc
from compression_amd import synthetic
from oracle import oracle
port = oracle.port()
dev = lambda a, dt=torch.int32: torch.as_tensor(np.ascontiguousarray(a)).to(dt).cuda()
pmfs, _ = synthetic.gaussian_pmfs(num_tables=16, octave=2.0)
cdfs = [port.pmf_to_quantized_cdf(p, 12) for p in pmfs]
lk = synthetic.assemble_lookup(cdfs, 12, overflow=True)
lt = torch.from_numpy(lk)
S, E = 8, 400
v = synthetic.sample_symbols(lk, S, E, seed=1)
chan = np.tile(np.arange(E) % 16, (S, 1)).astype(np.int32)
for name, idx in (("index = channel pattern", chan), ("index all 3", np.full((S, E), 3, np.int32))):
    val = v if name.startswith("index = ch") else synthetic.sample_symbols(lk[: 0] if False else lk, S, E, seed=1)
    if "all 3" in name:
        rows = synthetic.lookup_rows(lk)
        u = np.random.default_rng(0).integers(0, 4096, (S, E))
        val = (np.searchsorted(np.asarray(rows[3][1]), u, side="right") - 1).astype(np.int32)
    want = port.encode(lk, val, index=idx)[0]
    arr = np.empty(S, dtype=object)
    for i, x in enumerate(want):
        arr[i] = x
    hd = tfc.create_range_decoder(arr, lt, mode="throughput")
    hd, out = tfc.entropy_decode_index(hd, dev(idx), [E], torch.int32)
    out = out.cpu().numpy()
    fin = tfc.entropy_decode_finalize(hd).numpy()
    bad = np.argwhere(out != val)
    print(name, "differ", len(bad), bad[:4].tolist(), "fin", int(fin.sum()), flush=True)
    if len(bad):
        s0, e0 = bad[0]
        print("  got ", out[s0, max(0, e0 - 2): e0 + 10].tolist())
        print("  want", val[s0, max(0, e0 - 2): e0 + 10].tolist())
