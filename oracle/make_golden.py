"""TEST INFRASTRUCTURE — regenerates tests/golden/*.npz from the reference's own code
(oracle/_ref/libtfc_ref.so: cc/lib/range_coder.cc and the op kernel files
cc/kernels/{range_coder,pmf_to_cdf,quantization}_kernels.cc compiled verbatim from /root/reference
behind the shim headers in oracle/shim).

Run in the build container (needs /root/reference):  python oracle/make_golden.py
The vectors are committed so that the GPU box (no /root/reference) and the
CPU test tier can pin both the restated oracle and the HIP kernels to bytes the
reference itself produced.
"""
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle  # noqa: E402
from compression_amd import synthetic  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def mt19937_mod5(n):
    """std::mt19937(0)() % 5 sequence (numpy's MT19937 with seed 0 via init_genrand matches)."""
    # numpy's legacy seeding uses init_genrand(seed) == std::mt19937(seed).
    rs = np.random.RandomState(0)
    # RandomState.randint consumes differently; draw raw 32-bit outputs instead.
    raw = rs.bytes(4 * n)
    vals = np.frombuffer(raw, dtype="<u4")
    return (vals % 5).astype(np.int32)


# StochasticRound fixtures: inputs come from a formula (so only the outputs are stored) and cover four
# waves of the GPU kernel with a ragged tail.
STOCHASTIC_N = 50001
STOCHASTIC_CASES = {  # name: (dtype code, step_size, seed)
    "f32_step1": (0, 1.0, (123, 456)),
    "f32_step075": (0, 0.75, (7,)),
    "bf16_step1": (1, 1.0, tuple(range(1, 11))),
    "f16_step05": (2, 0.5, (-5, 2 ** 31 - 1)),
}


def stochastic_inputs(n, code):
    """Values in [-100, 100): float32 array (code 0) or (uint16 bit patterns, code) for bfloat16 / float16."""
    i = np.arange(n, dtype=np.uint64)
    x = (((i * np.uint64(2654435761)) % np.uint64(1 << 32)).astype(np.float64) / 2.0 ** 32 * 200.0 - 100.0)
    x = x.astype(np.float32)
    x[::97] = np.round(x[::97])          # exact integers stay put
    x[5::101] = np.floor(x[5::101]) + 0.5
    if code == 0:
        return x
    if code == 1:
        return ((x.view(np.uint32) >> 16).astype(np.uint16), 1)
    return (x.astype(np.float16).view(np.uint16), 2)


def unbounded_safe_limit(overflow_width):
    """Largest |value| (minus slack for offsets) the reference's UnboundedIndexRangeEncode handles without
    shifting a uint32 by 32 or more."""
    return (1 << (overflow_width * (31 // overflow_width) - 1)) - 64


def ops_roundtrip(ref, lookup, value, index):
    """Strings from the reference's OWN op kernels (CreateRangeEncoder -> EntropyEncode{Channel,Index} ->
    EntropyEncodeFinalize, range_coder_kernels.cc compiled verbatim behind oracle/shim), decoded back with
    its decoder ops; the restated drivers over the compiled core must give the same bytes."""
    streams, elems = value.shape
    strings = ref.ops_encode(lookup, [streams], [(value, index)])
    outs, ok = ref.ops_decode(lookup, strings, [streams], [([elems], index)])
    assert (outs[0] == value).all() and ok.all()
    again, blob, offs = ref.encode(lookup, value, index=index, threads=1)
    assert again == strings
    return strings, blob, offs


def pmf_cases():
    """[(pmf [rows, n] float32, precision)]: exactly symmetric Gaussian / Laplacian tables (tied pairs),
    flat and two-level tables (almost everything tied), random tables, over- and under-normalised."""
    from scipy.stats import laplace, norm
    rng = np.random.Generator(np.random.PCG64(2024))
    cases = []
    for dist in (norm, laplace):
        for scale in np.exp(np.linspace(np.log(0.11), np.log(64), 14)):
            half = int(np.ceil(scale * 6)) + 1
            x = np.arange(-half, half + 1)
            pmf = (dist.cdf((x + .5) / scale) - dist.cdf((x - .5) / scale)).astype(np.float32)
            cases.append((np.maximum(pmf, pmf[::-1])[None], 12 if len(cases) % 3 else 16))
    for n in (2, 16, 17, 33, 100, 257):
        cases.append((np.full((1, n), 1.0 / n, np.float32), 10))
        two = np.where(np.arange(n) % 3 == 0, 2.0, 1.0).astype(np.float32)
        cases.append(((two / two.sum())[None] * np.float32(1.3), 12))
        cases.append(((two / two.sum())[None] * np.float32(0.6), 12))
    for n in (5, 64, 300):
        w = rng.random((4, n)) ** 2 + 1e-4
        for s in (0.5, 1.0, 1.7):
            cases.append(((w / w.sum(-1, keepdims=True) * s).astype(np.float32), 12))
    return cases


def main():
    ref = oracle.reference()
    assert ref is not None, "oracle/_ref not built (needs /root/reference)"
    os.makedirs(GOLD, exist_ok=True)

    # ---- K1..K7: raw-call known answers --------------------
    kats = {}

    def add(name, lower, upper, precision):
        lower = np.asarray(lower, np.int32)
        upper = np.asarray(upper, np.int32)
        precision = np.broadcast_to(np.asarray(precision, np.int32), lower.shape).copy()
        data = ref.raw_encode(lower, upper, precision)
        kats[name + "_lower"] = lower
        kats[name + "_upper"] = upper
        kats[name + "_precision"] = precision
        kats[name + "_bytes"] = np.frombuffer(data, np.uint8).copy()
        return data

    def add_syms(name, cdf, precision, syms):
        cdf = np.asarray(cdf, np.int32)
        syms = np.asarray(syms, np.int32)
        kats[name + "_cdf"] = cdf
        kats[name + "_syms"] = syms
        return add(name, cdf[syms], cdf[syms + 1], precision)

    k1 = add("K1", [16], [18], 5)
    k2 = add("K2", [], [], 1)
    k3 = add_syms("K3", [0, 1, 2], 1, [int(c) for c in "10110010111000011010"])
    k4 = add_syms("K4", [0, 4000, 4050, 4090, 4096], 12, [int(c) for c in "0001002030000100"])
    k5 = add_syms("K5", [0, 1, 65535, 65536], 16, [int(c) for c in "111101121100221"])
    k6 = add_syms("K6", [0, 1, 2, 4096], 12, [0 if i % 7 == 3 else 2 for i in range(40)])
    k7 = add_syms("K7", [0, 100, 1000, 3000, 4000, 4096], 12, mt19937_mod5(100000))
    print("K1", k1.hex(), "K2", k2.hex(), "K3", k3.hex(), "K4", k4.hex())
    print("K5", k5.hex(), "K6", k6.hex(), "K7", len(k7), "bytes crc32=%08x" % zlib.crc32(k7))
    # The survey recorded these from the same compiled reference:
    assert k1.hex() == "80" and k2.hex() == "" and k3.hex() == "b2e1a0"
    assert k4.hex() == "eb99fb9f" and k5.hex() == "0004ffe9001dffe8fff0ffff01"
    assert k6.hex() == "005ffff6be4fdc0533"
    np.savez_compressed(os.path.join(GOLD, "kat_raw.npz"), **kats)

    # ---- multi-stream channel / index cases with escapes ------------------
    port = oracle.port()
    pmfs, minima = synthetic.gaussian_pmfs(num_tables=24, octave=3.0)   # sigma .25 .. 51
    cdfs = [port.pmf_to_quantized_cdf(p, 12) for p in pmfs]
    lookup = synthetic.assemble_lookup(cdfs, 12, overflow=True)
    val = synthetic.sample_symbols(lookup, streams=6, elems=1000, seed=11, escape_fraction=0.02)
    strings, blob, offs = ops_roundtrip(ref, lookup, val, None)
    rng = np.random.Generator(np.random.PCG64(5))
    idx = rng.integers(0, 24, size=val.shape).astype(np.int32)
    val_i = np.zeros_like(val)
    rows = synthetic.lookup_rows(lookup)
    for t, (sp, cdf) in enumerate(rows):
        m = idx == t
        u = rng.integers(0, 1 << 12, size=int(m.sum()))
        s = np.minimum(np.searchsorted(cdf, u, side="right") - 1, len(cdf) - 3)
        val_i[m] = np.maximum(s, 0)
    esc = rng.random(val.shape) < 0.02
    val_i = np.where(esc, rng.integers(-3000, 3000, size=val.shape), val_i).astype(np.int32)
    strings_i, blob_i, offs_i = ops_roundtrip(ref, lookup, val_i, idx)
    np.savez_compressed(
        os.path.join(GOLD, "streams_escape.npz"), lookup=lookup, value=val, blob=blob, offsets=offs,
        index=idx, value_indexed=val_i, blob_indexed=blob_i, offsets_indexed=offs_i)

    # ---- precision sweep, 2-D (matrix) lookup, no escapes -----------------
    sweep = {}
    for prec in (1, 2, 5, 8, 12, 16):
        rng = np.random.Generator(np.random.PCG64(100 + prec))
        nsym = min(1 << prec, 40)
        rowsM = []
        for _ in range(5):
            w = rng.random(nsym) ** 3 + 1e-3
            pmf = (w / w.sum()).astype(np.float32)
            rowsM.append(port.pmf_to_quantized_cdf(pmf, prec))
        width = nsym + 2
        mat = np.full((5, width), 1 << prec, np.int32)
        for r, c in enumerate(rowsM):
            mat[r, 0] = prec
            mat[r, 1:1 + len(c)] = c
        v = np.empty((3, 777), np.int32)
        for j in range(777):
            c = rowsM[j % 5]
            u = rng.integers(0, 1 << prec, size=3)
            v[:, j] = np.searchsorted(c, u, side="right") - 1
        strs, b, o = ops_roundtrip(ref, mat, v, None)
        sweep[f"p{prec}_lookup"] = mat
        sweep[f"p{prec}_value"] = v
        sweep[f"p{prec}_blob"] = b
        sweep[f"p{prec}_offsets"] = o
    np.savez_compressed(os.path.join(GOLD, "precision_sweep.npz"), **sweep)

    # ---- legacy RangeEncode with broadcasting -----------------------------
    leg = {}
    rng = np.random.Generator(np.random.PCG64(77))

    def hist_cdf(shape, m, prec):
        w = rng.random(shape + (m,)) + 0.05
        pmf = (w / w.sum(-1, keepdims=True)).astype(np.float32)
        return port.pmf_to_quantized_cdf(pmf, prec)

    cases = {
        "nobroadcast": ((4, 5, 6), (4, 5, 6)),
        "bcast1": ((4, 5, 6), (4, 1, 6)),
        "bcast2": ((3, 4, 5, 2), (1, 4, 1, 2)),
        "bcastall": ((7, 9), (1, 1)),
    }
    for name, (dshape, cshape) in cases.items():
        m, prec = 9, 10
        cdf = hist_cdf(cshape, m, prec)
        full = np.broadcast_to(cdf, dshape + (m + 1,))
        u = rng.integers(0, 1 << prec, size=dshape)
        data = (np.sum(full[..., 1:] <= u[..., None], axis=-1)).astype(np.int16)
        # the reference's RangeEncode / RangeDecode op kernels (range_coding_kernels.cc compiled verbatim);
        # the restated drivers over the compiled core must agree
        enc = ref.use_op_kernels().range_encode(data, cdf, prec)
        back = ref.use_op_kernels().range_decode(enc, dshape, cdf, prec)
        assert (back == data).all()
        assert enc == ref.range_encode(data, cdf, prec)
        leg[name + "_data"] = data
        leg[name + "_cdf"] = cdf
        leg[name + "_precision"] = np.int32(prec)
        leg[name + "_bytes"] = np.frombuffer(enc, np.uint8).copy()
    np.savez_compressed(os.path.join(GOLD, "legacy_broadcast.npz"), **leg)

    # ---- deprecated UnboundedIndexRangeEncode ---------------------------------
    # (drivers restated from unbounded_index_range_coding_kernels.cc over the compiled core)
    unb = {}
    rng = np.random.Generator(np.random.PCG64(78))
    rows, width, prec = 6, 14, 11
    cdf = np.zeros((rows, width), np.int32)
    size = np.zeros(rows, np.int32)
    for r in range(rows):
        n = int(rng.integers(3, width + 1))
        size[r] = n
        cuts = np.sort(rng.choice(np.arange(1, 1 << prec), n - 2, replace=False))
        cdf[r, :n] = np.concatenate([[0], cuts, [1 << prec]])
    offset = rng.integers(-6, 6, rows).astype(np.int32)
    index = rng.integers(0, rows, (5, 41)).astype(np.int32)
    data = rng.integers(-30, 45, (5, 41)).astype(np.int32)
    data[0, :4] = [-70000, 70000, 2 ** 30, -(2 ** 30)]          # long overflow codes
    unb.update(cdf=cdf, cdf_size=size, offset=offset, index=index, data=data, precision=np.int32(prec))
    ops = ref.use_op_kernels()   # unbounded_index_range_coding_kernels.cc compiled verbatim
    for ow in (1, 2, 4, 7, 16):
        enc = ref.unbounded_index_range_encode(data, index, cdf, size, offset, prec, ow)
        assert (ref.unbounded_index_range_decode(enc, index, cdf, size, offset, prec, ow) == data).all()
        unb[f"w{ow}_bytes"] = np.frombuffer(enc, np.uint8).copy()
        # The reference op counts digits with `overflow >> (widths * overflow_width)` (:229), which shifts
        # a uint32 by >= 32 (undefined; an endless loop on x86) once the overflow needs more than
        # floor(31 / width) digits.  Inside that domain the op itself produces the golden bytes; the
        # full-range bytes above come from the restated driver, which counts in 64 bits.
        safe = np.clip(data, -unbounded_safe_limit(ow), unbounded_safe_limit(ow))
        enc = ops.unbounded_index_range_encode(safe, index, cdf, size, offset, prec, ow)
        assert enc == ref.unbounded_index_range_encode(safe, index, cdf, size, offset, prec, ow)
        assert (ops.unbounded_index_range_decode(enc, index, cdf, size, offset, prec, ow) == safe).all()
        unb[f"w{ow}_safe_bytes"] = np.frombuffer(enc, np.uint8).copy()
    np.savez_compressed(os.path.join(GOLD, "unbounded_index.npz"), **unb)
    # ---- PmfToQuantizedCdf (the reference kernel file itself, compiled behind oracle/shim) ----
    # Tie-heavy on purpose: which of two equal-penalty symbols is adjusted depends on std::sort's order.
    pc = {}
    for k, (pmf, prec) in enumerate(pmf_cases()):
        pc[f"pmf{k}"] = pmf
        pc[f"precision{k}"] = np.int32(prec)
        pc[f"cdf{k}"] = ref.pmf_to_quantized_cdf(pmf, prec)
    pc["count"] = np.int32(k + 1)
    np.savez_compressed(os.path.join(GOLD, "pmf_to_cdf.npz"), **pc)

    # ---- StochasticRound (the reference kernel file itself, compiled behind oracle/shim) ----
    sr = {}
    for name, (code, step, seed) in STOCHASTIC_CASES.items():
        out = ref.stochastic_round(stochastic_inputs(STOCHASTIC_N, code), step, seed)
        assert np.abs(out).max() < 2 ** 15
        sr[name] = out.astype(np.int16)
    np.savez_compressed(os.path.join(GOLD, "stochastic_round.npz"), **sr)
    print("golden vectors written to", GOLD)


if __name__ == "__main__":
    main()
