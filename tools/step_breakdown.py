"""Host-side timing of each op of one bench step (synchronised after every call)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import compression_amd as tfc
from compression_amd import synthetic

dev = torch.device("cuda")
lookup = bench.build_tables(dev)
value = synthetic.sample_symbols(lookup, bench.STREAMS, bench.ELEMS, seed=0)
lt, vt = torch.from_numpy(lookup), torch.from_numpy(value).to(dev)
for _ in range(3):
    bench.one_step(lt, vt)
acc = {}
def T(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    acc[name] = acc.get(name, 0) + time.perf_counter() - t0; return r
N = 10
for _ in range(N):
    h = T("create_range_encoder", lambda: tfc.create_range_encoder([bench.STREAMS], lt))
    h = T("entropy_encode_channel", lambda: tfc.entropy_encode_channel(h, vt))
    blob, offs = T("finalize+read", lambda: tfc.gen_ops._finalize_device(h))
    d = T("create_range_decoder", lambda: tfc.create_range_decoder((blob, offs, (bench.STREAMS,)), lt))
    d, out = T("entropy_decode_channel", lambda: tfc.entropy_decode_channel(d, [bench.ELEMS], torch.int32))
    ok = T("entropy_decode_finalize", lambda: tfc.entropy_decode_finalize(d))
for k, v in acc.items():
    print(f"{k:28s} {1e3 * v / N:8.3f} ms")
print("sum", 1e3 * sum(acc.values()) / N)
