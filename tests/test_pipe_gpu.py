"""GPU tier: the pipelined lane kernels (csrc/range_pipe.h: parallel expansion into call words + chain for
encode, chain into raw rows + parallel parse for decode) that throughput-mode handles run.  Bytes and symbols must
equal the oracle's (cc/kernels/range_coder_kernels.cc:191-322, 360-471; cc/lib/range_coder.cc:37-307) — and it must
be THESE kernels that produced them: tfc_pipe_counters tells whether the lane-per-stream fallback had to take a job.
(tests/test_range_coder_gpu.py runs every golden vector through the same kernels via the "throughput" fixture.)"""
import ctypes as C

import numpy as np
import pytest
import torch

from compression_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tfc():
    import compression_amd
    compression_amd.set_default_mode("throughput")
    yield compression_amd
    compression_amd.set_default_mode("auto")


def counters():
    from compression_amd import _lib
    a, b = C.c_int64(), C.c_int64()
    _lib.check(_lib.lib().tfc_pipe_counters(C.byref(a), C.byref(b)))
    return a.value, b.value


def dev(a, dtype=torch.int32):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).cuda()


def tables(port, n, prec=12, overflow=True, octave=2.0):
    pmfs, _ = synthetic.gaussian_pmfs(num_tables=n, octave=octave)
    return synthetic.assemble_lookup([port.pmf_to_quantized_cdf(p, prec) for p in pmfs], prec, overflow=overflow)


def roundtrip(tfc, port, lookup, value, index=None, calls=1):
    """encode -> bytes == oracle; decode(oracle bytes) == value; Finalize flags true.  -> (pipelined launches,
    fallback workgroups) of the calls."""
    streams, elems = value.shape
    lt = torch.from_numpy(lookup)
    want = port.encode(lookup, value, index=index, calls=calls)[0]
    l0, f0 = counters()
    h = tfc.create_range_encoder([streams], lt)
    bounds = [elems * k // calls for k in range(calls + 1)]
    for a, b in zip(bounds[:-1], bounds[1:]):
        if index is None:
            h = tfc.entropy_encode_channel(h, dev(value[:, a:b]))
        else:
            h = tfc.entropy_encode_index(h, dev(index[:, a:b]), dev(value[:, a:b]))
    got = [bytes(x) for x in tfc.entropy_encode_finalize(h).reshape(-1)]
    assert got == want
    arr = np.empty(len(want), dtype=object)
    for i, x in enumerate(want):
        arr[i] = x
    hd = tfc.create_range_decoder(arr, lt)
    outs = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        if index is None:
            hd, o = tfc.entropy_decode_channel(hd, [b - a], torch.int32)
        else:
            hd, o = tfc.entropy_decode_index(hd, dev(index[:, a:b]), [b - a], torch.int32)
        outs.append(o.cpu().numpy())
    assert (np.concatenate(outs, axis=1) == value).all()
    assert bool(tfc.entropy_decode_finalize(hd).all())
    l1, f1 = counters()
    return l1 - l0, f1 - f0


@pytest.mark.parametrize("streams,elems", [(8, 2048), (70, 1000), (64, 256), (1, 17), (130, 513), (3, 4099)])
def test_channel_mode_geometries(tfc, port, streams, elems):
    """Stream counts around the 64-stream groups, element counts around the 256-symbol tiles and 16-row blocks."""
    lookup = tables(port, 16)
    launches, fallback = roundtrip(tfc, port, lookup, synthetic.sample_symbols(lookup, streams, elems, seed=streams))
    assert launches == 2 and fallback == 0


def test_escape_codes_are_rows(tfc, port):
    """1 % of the symbols far outside their tables (second run): the Elias-gamma bits are ordinary rows
    of the chain (range_coder_kernels.cc:304-321, 449-471) — also across calls on one handle."""
    lookup = tables(port, 16)
    value = synthetic.sample_symbols(lookup, 96, 3072, seed=1, escape_fraction=0.01)
    assert roundtrip(tfc, port, lookup, value) == (2, 0)
    assert roundtrip(tfc, port, lookup, value, calls=3) == (6, 0)      # (call lengths: multiples of the table count)


def test_config2_tables(tfc, port):
    """The 192 tables of BASELINE config 2 (149 KB decoder image: one workgroup per CU), escape symbols in the data."""
    lookup = tables(port, 192, octave=24.0)
    value = synthetic.sample_symbols(lookup, 512, 4096, seed=4, escape_fraction=0.004)
    assert roundtrip(tfc, port, lookup, value) == (2, 0)


def test_tables_without_escape_rows_and_few_tables(tfc, port):
    """Positive precision headers (no escape symbol; exactly one row per symbol) and fewer tables than rows in a
    block (the channel cursor wraps several times per block)."""
    lookup = tables(port, 8, overflow=False)
    assert roundtrip(tfc, port, lookup, synthetic.sample_symbols(lookup, 40, 1111, seed=3)) == (2, 0)


def test_index_mode(tfc, port):
    """EntropyEncodeIndex / EntropyDecodeIndex (range_coder_kernels.cc:217-242, 380-404): the table of every element
    from the index tensor; values drawn through the indexed table, some of them far outside it."""
    lookup = tables(port, 16)
    rng = np.random.default_rng(5)
    index = rng.integers(0, 16, (80, 1800)).astype(np.int32)
    rows = synthetic.lookup_rows(lookup)
    u = rng.integers(0, 1 << 12, index.shape)
    value = np.zeros(index.shape, np.int32)
    for t, (_, c) in enumerate(rows):
        m = index == t
        value[m] = np.searchsorted(np.asarray(c), u[m], side="right") - 1
    assert roundtrip(tfc, port, lookup, value, index=index) == (2, 0)
    wild = np.where(rng.random(index.shape) < 0.005, rng.integers(-300, 300, index.shape) * 7, value).astype(np.int32)
    assert roundtrip(tfc, port, lookup, wild, index=index) == (2, 0)
    assert roundtrip(tfc, port, lookup, wild, index=index, calls=3) == (6, 0)


def test_more_rows_than_planned_fall_back(tfc, port):
    """Every symbol an escape with a long code: far more coder calls than the launch plans rows for.  The pipelined
    kernels leave such a job untouched and the lane-per-stream kernel behind them codes it — same bytes."""
    lookup = tables(port, 16)
    rng = np.random.default_rng(21)
    big = (rng.integers(1 << 12, 1 << 30, (4, 333)) * rng.choice([-1, 1], (4, 333))).astype(np.int32)
    launches, fallback = roundtrip(tfc, port, lookup, big)
    assert launches >= 2 and fallback >= 2      # (the encode call is repeated once with the worst-case slab)


def test_precision_16_one_digit_per_symbol(tfc, port):
    """Precision-16 rows whose plain symbols have probability 2^-16 (a 16-bit digit per symbol: the encoder's slab is
    outgrown and the call repeated with the bound that cannot be), mixed with escape codes."""
    cdf = list(range(0, 9)) + [65535, 65536]
    lookup = np.array([[-16] + cdf, [-16] + cdf], np.int32)
    rng = np.random.default_rng(77)
    u = rng.random((70, 900))
    value = np.where(u < 0.03, rng.integers(9, 2000, (70, 900)), np.where(u < 0.6, 8, rng.integers(0, 8, (70, 900)))).astype(np.int32)
    roundtrip(tfc, port, lookup, value)


def test_many_handles_in_one_launch(tfc, port):
    """tfc_encoder_encode_many / tfc_decoder_decode_many: several 512-stream batches as one launch per stage."""
    lookup = tables(port, 16)
    lt = torch.from_numpy(lookup)
    values = [synthetic.sample_symbols(lookup, 130, 700, seed=40 + k, escape_fraction=0.01) for k in range(5)]
    l0, f0 = counters()
    hs = tfc.create_range_encoders(5, [130], lt, deferred_errors=True)
    hs = tfc.entropy_encode_channel_many(hs, [dev(v) for v in values])
    hs = tfc.entropy_encode_finalize_device_many(hs)
    ds = tfc.create_range_decoders(hs, lt)
    ds, decoded = tfc.entropy_decode_channel_many(ds, [700], torch.int32)
    oks = tfc.entropy_decode_finalize_device_many(ds)
    for h, v, d, ok in zip(hs, values, decoded, oks):
        assert [bytes(s) for s in tfc.fetch_strings(h)] == port.encode(lookup, v)[0]
        assert (d.cpu().numpy().reshape(130, 700) == v).all() and bool(ok.all())
    l1, f1 = counters()
    assert (l1 - l0, f1 - f0) == (2, 0)


@pytest.mark.parametrize("streams,elems", [(4096, 700), (4100, 333)])
def test_large_launches_take_the_overlapped_path(tfc, port, streams, elems):
    """64 groups and more: the encoder's chain workgroups (a chain wave, two loader waves and a storer wave per group)
    run on the library's own stream beside the expansion, the decoder's parse beside its chain with a second pass
    behind it (csrc/range_pipe.h).  Same bytes, same symbols, no fallback — with escape codes in the data, a last
    group that is not full and element counts that end inside a tile, an iteration and a block."""
    lookup = tables(port, 16)
    value = synthetic.sample_symbols(lookup, streams, elems, seed=streams, escape_fraction=0.01)
    assert roundtrip(tfc, port, lookup, value) == (2, 0)


def test_large_launch_of_many_handles_in_index_mode(tfc, port):
    """Ten 512-stream handles behind one launch per stage (80 groups), EntropyEncodeIndex / EntropyDecodeIndex with
    values far outside the tables in between."""
    lookup = tables(port, 16)
    lt = torch.from_numpy(lookup)
    rng = np.random.default_rng(9)
    rows = synthetic.lookup_rows(lookup)
    values, indexes = [], []
    for k in range(10):
        index = rng.integers(0, 16, (512, 520)).astype(np.int32)
        u = rng.integers(0, 1 << 12, index.shape)
        value = np.zeros(index.shape, np.int32)
        for t, (_, c) in enumerate(rows):
            m = index == t
            value[m] = np.searchsorted(np.asarray(c), u[m], side="right") - 1
        value = np.where(rng.random(index.shape) < 0.004, rng.integers(-300, 300, index.shape) * 5, value).astype(np.int32)
        values.append(value)
        indexes.append(index)
    from compression_amd import _lib
    l0, f0 = counters()
    hs = tfc.create_range_encoders(10, [512], lt, deferred_errors=True)
    keep = [(dev(i), dev(v)) for i, v in zip(indexes, values)]
    hp = (C.c_void_p * 10)(*[h.ptr for h in hs])
    ip = (C.c_void_p * 10)(*[i.data_ptr() for i, _ in keep])
    vp = (C.c_void_p * 10)(*[v.data_ptr() for _, v in keep])
    _lib.check(_lib.lib().tfc_encoder_encode_many(10, hp, vp, ip, 520, _lib.stream_ptr()))
    hs = tfc.entropy_encode_finalize_device_many(hs)
    ds = tfc.create_range_decoders(hs, lt)
    outs = [torch.empty(512, 520, dtype=torch.int32, device="cuda") for _ in range(10)]
    dp = (C.c_void_p * 10)(*[d.ptr for d in ds])
    op = (C.c_void_p * 10)(*[o.data_ptr() for o in outs])
    _lib.check(_lib.lib().tfc_decoder_decode_many(10, dp, ip, op, 520, _lib.stream_ptr()))
    oks = tfc.entropy_decode_finalize_device_many(ds)
    for h, v, i, o, ok in zip(hs, values, indexes, outs, oks):
        assert [bytes(s) for s in tfc.fetch_strings(h)] == port.encode(lookup, v, index=i)[0]
        assert (o.cpu().numpy() == v).all() and bool(ok.all())
    l1, f1 = counters()
    assert (l1 - l0, f1 - f0) == (2, 0), (l1 - l0, f1 - f0)


def test_more_groups_than_the_chip_hosts_at_once(tfc, port):
    """40 handles of 512 streams behind one call per direction = 320 groups: a launch takes only as many batches as its
    chain workgroups can be resident at once (an encoder chain workgroup per 2 groups on at most half the CUs, a decoder
    chain workgroup per CU at most), so the call is several launches — and none of them leaves a job to the fallback.
    (Round 5: with two rounds of chain workgroups in ONE launch the parse next to the chain was abandoned mid-way, and
    workgroups that started at that moment went on with some of their threads: wrong elements for 16-64 streams of a
    tile.)  Every batch's bytes against the oracle, every element back."""
    lookup = tables(port, 16)
    lt = torch.from_numpy(lookup)
    n = 40
    values = [synthetic.sample_symbols(lookup, 512, 600, seed=900 + k, escape_fraction=0.01) for k in range(n)]
    l0, f0 = counters()
    hs = tfc.create_range_encoders(n, [512], lt, deferred_errors=True)
    hs = tfc.entropy_encode_channel_many(hs, [dev(v) for v in values])
    hs = tfc.entropy_encode_finalize_device_many(hs)
    ds = tfc.create_range_decoders(hs, lt)
    ds, decoded = tfc.entropy_decode_channel_many(ds, [600], torch.int32)
    oks = tfc.entropy_decode_finalize_device_many(ds)
    for k, (h, v, d, ok) in enumerate(zip(hs, values, decoded, oks)):
        assert (d.cpu().numpy().reshape(512, 600) == v).all() and bool(ok.all()), k
        if k % 8 == 0:
            assert [bytes(s) for s in tfc.fetch_strings(h)] == port.encode(lookup, v)[0], k
    l1, f1 = counters()
    assert l1 - l0 >= 3 and f1 - f0 == 0, (l1 - l0, f1 - f0)
