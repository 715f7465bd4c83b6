"""Runs one sub-case of tests/test_range_coder_gpu.py::test_small_shapes_both_modes (to localise a hang)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import compression_amd as tfc
from compression_amd import synthetic
from oracle import oracle
import test_range_coder_gpu as T
port = oracle.port()
case, mode = sys.argv[1], sys.argv[2]
tfc.set_default_mode(mode)
lookup = np.load(os.path.join(T.__file__.rsplit("/", 1)[0], "golden", "streams_escape.npz"))["lookup"]
rows = synthetic.lookup_rows(lookup)
rng = np.random.default_rng(33)
if case.startswith("shape"):
    streams, elems = [int(x) for x in case[5:].split("x")]
    value = synthetic.sample_symbols(lookup, streams, elems, seed=streams * 1000 + elems)
    want = port.encode(lookup, value)[0]
    print("enc", T.hip_encode(tfc, lookup, value)[0] == want, flush=True)
    d, ok = T.hip_decode(tfc, lookup, want, elems)
    print("dec", (d == value).all(), ok.all(), flush=True)
    index = rng.integers(0, len(rows), value.shape).astype(np.int32)
    vi = np.zeros_like(value)
    for t, (sp, cdf) in enumerate(rows):
        m = index == t
        vi[m] = rng.integers(0, len(cdf) - 2, int(m.sum()))
    want = port.encode(lookup, vi, index=index)[0]
    print("enc idx", T.hip_encode(tfc, lookup, vi, index=index)[0] == want, flush=True)
    d, ok = T.hip_decode(tfc, lookup, want, elems, index=index)
    print("dec idx", (d == vi).all(), ok.all(), flush=True)
elif case == "extreme":
    value = np.array([[2**30 - 1, -(2**30 - 1), 2**29, -(2**29), 5, -1, 0, 1]], np.int32)
    want = port.encode(lookup, value)[0]
    print("enc", T.hip_encode(tfc, lookup, value)[0] == want, flush=True)
    d, ok = T.hip_decode(tfc, lookup, want, value.shape[1])
    print("dec", (d == value).all(), ok.all(), flush=True)
elif case == "multidec":
    v = synthetic.sample_symbols(lookup, 6, 1440, seed=1, escape_fraction=0.02)
    want = port.encode(lookup, v)[0]
    arr = np.empty(6, dtype=object)
    for i, x in enumerate(want):
        arr[i] = x
    hd = tfc.create_range_decoder(arr, torch.as_tensor(lookup))
    parts = []
    for _ in range(3):
        hd, out = tfc.entropy_decode_channel(hd, [480], torch.int32)
        torch.cuda.synchronize()
        print("call done", flush=True)
        parts.append(out.cpu().numpy())
    print("dec", (np.concatenate(parts, axis=1) == v).all(), bool(tfc.entropy_decode_finalize(hd).all()), flush=True)
elif case == "zerowidth":
    lookup = np.array([[-8, 0, 0, 100, 100, 200, 256, 256, 256],
                       [8, 0, 64, 64, 64, 128, 256, 256, 256]], np.int32)
    rng = np.random.default_rng(4)
    value = np.empty((5, 400), np.int32)
    value[:, 0::2] = rng.choice([1, 3, 4, 9, -3], (5, 200))
    value[:, 1::2] = rng.choice([0, 3, 4], (5, 200))
    want = port.encode(lookup, value)[0]
    print("enc", T.hip_encode(tfc, lookup, value)[0] == want, flush=True)
    d, ok = T.hip_decode(tfc, lookup, want, 400)
    print("dec", (d == value).all(), ok.all(), flush=True)
