import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
import importlib.util
spec = importlib.util.spec_from_file_location("fz", "tests/test_pipe_fuzz_gpu.py"); fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
import compression_amd as tfc
from oracle import oracle
tfc.set_default_mode("throughput")
port = oracle.port()
for (precision, ntab, streams, elems, indexed, esc, seed) in fz.CASES:
    rng = np.random.default_rng(seed)
    lookup = fz.random_lookup(port, rng, ntab, precision)
    index = rng.integers(0, ntab, (streams, elems)).astype(np.int32) if indexed else None
    value = fz.random_values(rng, lookup, index, streams, elems, esc)
    lt = torch.from_numpy(lookup)
    l0, f0 = fz.counters()
    h = tfc.create_range_encoder([streams], lt)
    h = tfc.entropy_encode_channel(h, fz.dev(value)) if index is None else tfc.entropy_encode_index(h, fz.dev(index), fz.dev(value))
    got = tfc.entropy_encode_finalize(h)
    l1, f1 = fz.counters()
    hd = tfc.create_range_decoder(got, lt)
    if index is None: hd, out = tfc.entropy_decode_channel(hd, [elems], torch.int32)
    else: hd, out = tfc.entropy_decode_index(hd, fz.dev(index), [elems], torch.int32)
    torch.cuda.synchronize()
    l2, f2 = fz.counters()
    nesc = int(((value < 0) | (value > 400)).sum())
    print("p", precision, "ntab", ntab, "s", streams, "e", elems, "ix", int(indexed), "esc", esc, "| enc launches", l1 - l0, "fallback", f1 - f0, "| dec launches", l2 - l1, "fallback", f2 - f1, "| far values", nesc, "of", value.size)
