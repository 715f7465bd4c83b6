"""The pipelined lane kernels (csrc/range_pipe.h) against the oracle, case by case, without stopping at the
first failure.  Run with TFC_PIPE_NOFALLBACK=1 to see jobs the pipelined kernels gave up on as failures
(the lane-per-stream kernel behind them is then not launched)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import compression_amd as tfc
from compression_amd import synthetic
from oracle import oracle

port = oracle.port()
dev = lambda a, dt=torch.int32: torch.as_tensor(np.ascontiguousarray(a)).to(dt).cuda()
fails = []


def tables(n, prec=12, overflow=True, octave=2.0):
    pmfs, _ = synthetic.gaussian_pmfs(num_tables=n, octave=octave)
    cdfs = [port.pmf_to_quantized_cdf(p, prec) for p in pmfs]
    return synthetic.assemble_lookup(cdfs, prec, overflow=overflow)


def case(name, lookup, value, index=None, calls=1, encode=True, decode=True):
    streams, elems = value.shape
    lt = torch.from_numpy(lookup)
    want = port.encode(lookup, value, index=index, calls=calls)[0]
    ok_e = ok_d = None
    try:
        if encode:
            h = tfc.create_range_encoder([streams], lt, mode="throughput")
            bounds = [elems * k // calls for k in range(calls + 1)]
            for a, b in zip(bounds[:-1], bounds[1:]):
                if index is None:
                    h = tfc.entropy_encode_channel(h, dev(value[:, a:b]))
                else:
                    h = tfc.entropy_encode_index(h, dev(index[:, a:b]), dev(value[:, a:b]))
            got = [bytes(x) for x in tfc.entropy_encode_finalize(h).reshape(-1)]
            ok_e = got == want
            if not ok_e:
                bad = [i for i, (g, w) in enumerate(zip(got, want)) if g != w]
                first = bad[0]
                g, w = got[first], want[first]
                at = next((i for i in range(min(len(g), len(w))) if g[i] != w[i]), min(len(g), len(w)))
                print(f"   encode: {len(bad)}/{streams} streams differ; stream {first}: len {len(g)} vs {len(w)}, first byte {at}")
        if decode:
            arr = np.empty(len(want), dtype=object)
            for i, x in enumerate(want):
                arr[i] = x
            hd = tfc.create_range_decoder(arr, lt, mode="throughput")
            if calls == 1:
                if index is None:
                    hd, out = tfc.entropy_decode_channel(hd, [elems], torch.int32)
                else:
                    hd, out = tfc.entropy_decode_index(hd, dev(index), [elems], torch.int32)
                out = out.cpu().numpy()
            else:
                outs = []
                bounds = [elems * k // calls for k in range(calls + 1)]
                for a, b in zip(bounds[:-1], bounds[1:]):
                    if index is None:
                        hd, o = tfc.entropy_decode_channel(hd, [b - a], torch.int32)
                    else:
                        hd, o = tfc.entropy_decode_index(hd, dev(index[:, a:b]), [b - a], torch.int32)
                    outs.append(o.cpu().numpy())
                out = np.concatenate(outs, axis=1)
            fin = tfc.entropy_decode_finalize(hd).numpy()
            ok_d = bool((out == value).all() and fin.all())
            if not ok_d:
                bad = np.argwhere(out != value)
                print(f"   decode: {len(bad)} elements differ, first {bad[:3].tolist()}, finalize ok {int(fin.sum())}/{streams}")
    except Exception as e:  # noqa: BLE001
        print("   exception:", repr(e)[:300])
        ok_e = False
    good = (ok_e is not False) and (ok_d is not False)
    print(("PASS " if good else "FAIL ") + name, flush=True)
    if not good:
        fails.append(name)


rng = np.random.default_rng(5)
if "--time-only" in sys.argv:
    case = lambda *a, **k: None
lk16 = tables(16)
lk192 = tables(192, octave=24.0)
lk_noesc = tables(8, overflow=False)
for streams, elems in ((8, 2048), (70, 1000), (64, 256), (1, 17), (130, 513), (3, 4099)):
    v = synthetic.sample_symbols(lk16, streams, elems, seed=streams)
    case(f"channel natural {streams}x{elems}", lk16, v)
v = synthetic.sample_symbols(lk16, 96, 3072, seed=1, escape_fraction=0.01)
case("channel 1% escapes 96x3072", lk16, v)
case("channel 1% escapes, 3 calls", lk16, v, calls=3)       # (call lengths: multiples of the table count)
nofb = os.environ.get("TFC_PIPE_NOFALLBACK", "0") not in ("", "0")
if not nofb:
    case("channel 1% escapes, table phase broken by the split (fallback)", lk16, v[:, :3000], calls=3)
v = synthetic.sample_symbols(lk16, 64, 1500, seed=2, escape_fraction=0.05)
if not nofb:
    case("channel 5% escapes 64x1500 (more rows than planned: fallback)", lk16, v)
v = synthetic.sample_symbols(lk_noesc, 40, 1111, seed=3)
case("no escape rows 40x1111", lk_noesc, v)
v = synthetic.sample_symbols(lk192, 512, 4096, seed=4, escape_fraction=0.004)
case("C2 tables 512x4096 0.4%", lk192, v)
# index mode: every element carries its table
idx = rng.integers(0, 16, (80, 1800)).astype(np.int32)
pmfs, _ = synthetic.gaussian_pmfs(num_tables=16, octave=2.0)
rows = synthetic.lookup_rows(lk16)
u = rng.integers(0, 1 << 12, idx.shape)
vi = np.zeros(idx.shape, np.int32)
for t, (_, c) in enumerate(rows):
    m = idx == t
    vi[m] = np.searchsorted(np.asarray(c), u[m], side="right") - 1
case("index natural 80x1800", lk16, vi, index=idx)
esc = rng.random(idx.shape) < 0.005
vi2 = np.where(esc, rng.integers(-300, 300, idx.shape) * 7, vi).astype(np.int32)
case("index 0.5% wild values 80x1800", lk16, vi2, index=idx)
case("index, 2 calls", lk16, vi2, index=idx, calls=2)
case("index, 3 calls", lk16, vi2, index=idx, calls=3)
if not nofb:
    vi3 = np.where(rng.random(idx.shape) < 0.05, rng.integers(-300, 300, idx.shape) * 7, vi).astype(np.int32)
    case("index 5% wild values (fallback)", lk16, vi3, index=idx)
# dense long escapes: more rows than the pipelined kernels plan for -> their fallback
big = (rng.integers(1 << 12, 1 << 30, (4, 333)) * rng.choice([-1, 1], (4, 333))).astype(np.int32)
if os.environ.get("TFC_PIPE_NOFALLBACK", "0") in ("", "0"):
    case("dense long escapes 4x333 (fallback)", lk16, big)
# precision 16 with width-1 symbols (one digit per symbol) and escapes
cdf = list(range(0, 9)) + [65535, 65536]
lk_p16 = np.array([[-16] + cdf, [-16] + cdf], np.int32)
rare = rng.integers(0, 8, (70, 900))
uu = rng.random((70, 900))
v16 = np.where(uu < 0.03, rng.integers(9, 2000, (70, 900)), np.where(uu < 0.6, 8, rare)).astype(np.int32)
case("precision 16 rare symbols 70x900", lk_p16, v16)

# fused quantise / dequantise
C = 16
lt = torch.from_numpy(lk16)
y = (torch.randn(32, 40, C, device="cuda") * 3).to(torch.bfloat16)
from compression_amd.entropy_models import continuous_batched  # noqa: E402,F401
print("failures:", fails)

# timing: config 2, 20 batches per launch
if "--time" in sys.argv or "--time-only" in sys.argv:
    lt192 = torch.from_numpy(lk192)
    for frac, law in ((0.0, "natural 0.4%"), (0.01, "1% geometric")):
        vals = [dev(synthetic.sample_symbols(lk192, 512, 49152, seed=10 + k, escape_fraction=frac)) for k in range(2)]
        vals = [vals[k % 2] for k in range(20)]
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            hs = tfc.create_range_encoders(20, [512], lt192, mode="throughput", deferred_errors=True)
            hs = tfc.entropy_encode_channel_many(hs, vals)
            hs = tfc.entropy_encode_finalize_device_many(hs)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ds = tfc.create_range_decoders(hs, lt192, mode="throughput")
            ds, dec = tfc.entropy_decode_channel_many(ds, [49152], torch.int32)
            oks = tfc.entropy_decode_finalize_device_many(ds)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            good = all(torch.equal(d.reshape(512, 49152), v) for d, v in zip(dec, vals)) and all(bool(o.all()) for o in oks)
            print(f"{law}: encode {1e3 * (t1 - t0):.2f} ms, decode {1e3 * (t2 - t1):.2f} ms per 20 batches, round trip exact {good}", flush=True)
            del hs, ds, dec, oks
