"""GPU tier: the stream-ordered (nothing read back) coding path and model steps in flight on several streams.

What must hold: a deferred-error handle of the wave-per-stream family codes without a host round trip and
still produces the oracle's bytes; range errors and an outgrown speculative slab surface at status /
fetch; `compress(device_result=True)` + `decompress(defer_sanity=True)` on a lane of pipeline.StepLanes give
the strings and images of the plain calls; several steps in flight do not disturb each other."""
import os

import numpy as np
import pytest
import torch

import compression_amd as tfc
from compression_amd import pipeline, synthetic

pytestmark = pytest.mark.gpu


def _tables(num=16, escape=True):
    from oracle import oracle
    port = oracle.port()
    pmfs, _ = synthetic.gaussian_pmfs(num_tables=num, octave=2.0)
    cdfs = [port.pmf_to_quantized_cdf(p, 12) for p in pmfs]
    return port, synthetic.assemble_lookup(cdfs, 12, overflow=escape)


@pytest.mark.parametrize("mode", ["latency", "throughput"])
@pytest.mark.parametrize("indexed", [False, True])
def test_deferred_handle_codes_without_readback_and_matches_oracle(mode, indexed):
    port, lookup = _tables()
    value = synthetic.sample_symbols(lookup, 24, 5000, seed=3, escape_fraction=0.01)
    index = None
    if indexed:
        index = np.random.default_rng(1).integers(0, 16, size=value.shape).astype(np.int32)
        value = np.minimum(value, 3).astype(np.int32)       # in range of (or an escape for) every table
    want, _, _ = port.encode(lookup, value, index=index)
    lt = torch.from_numpy(lookup)
    h = tfc.create_range_encoder([24], lt, mode=mode, deferred_errors=True)
    v = torch.from_numpy(value).cuda()
    if indexed:
        i = torch.from_numpy(index).cuda()
        h = tfc.entropy_encode_index(h, i[:, :2000].contiguous(), v[:, :2000].contiguous())
        h = tfc.entropy_encode_index(h, i[:, 2000:].contiguous(), v[:, 2000:].contiguous())
    else:
        h = tfc.entropy_encode_channel(h, v)
    h = tfc.entropy_encode_finalize_device(h)
    blob, offsets = tfc.device_strings(h)
    d = tfc.create_range_decoder(h, lt, mode=mode)
    if indexed:
        d, dec = tfc.entropy_decode_index(d, torch.from_numpy(index).cuda(), [5000], torch.int32)
    else:
        d, dec = tfc.entropy_decode_channel(d, [5000], torch.int32)
    ok = tfc.entropy_decode_finalize_device(d)
    # only now does the host look at anything
    got = tfc.fetch_strings(h)
    assert [bytes(s) for s in got] == want
    off = offsets.cpu().numpy()
    assert int(off[-1]) == sum(map(len, want)) and blob.numel() >= int(off[-1])
    assert bytes(blob[:int(off[-1])].cpu().numpy().tobytes()) == b"".join(want)
    assert torch.equal(dec.cpu(), torch.from_numpy(value)) and bool(ok.cpu().all())


def test_deferred_range_error_is_reported_by_status():
    port, lookup = _tables(escape=False)
    value = synthetic.sample_symbols(lookup, 8, 1000, seed=4)
    value[5, 777] = 10 ** 6
    h = tfc.create_range_encoder([8], torch.from_numpy(lookup), mode="latency", deferred_errors=True)
    h = tfc.entropy_encode_channel(h, torch.from_numpy(value).cuda())       # returns without looking
    h = tfc.entropy_encode_finalize_device(h)
    with pytest.raises(ValueError, match=r"value=1000000 not in range \[0, "):
        tfc.fetch_strings(h)


def test_outgrown_speculative_slab_is_coded_again_not_an_error():
    """A deferred wave-per-stream call sizes its slab without the counting pass's result; data that needs more (here:
    the slab shrunk by the test hook) must never write past the slab — and must not fail either: the reference codes
    any encodable input.  The C ABI reports the flagged handle (tfc_encoder_status); the op layer then issues the
    handle's encode calls again on a synchronising encoder and returns the oracle's bytes."""
    from compression_amd import _lib
    port, lookup = _tables()
    value = synthetic.sample_symbols(lookup, 8, 20000, seed=5, escape_fraction=0.01)
    want, _, _ = port.encode(lookup, value)
    os.environ["TFC_SPECULATIVE_SLAB_DIV"] = "1000"
    try:
        # the library's own verdict
        h = tfc.create_range_encoder([8], torch.from_numpy(lookup), mode="latency", deferred_errors=True)
        h = tfc.entropy_encode_channel(h, torch.from_numpy(value).cuda())
        h = tfc.entropy_encode_finalize_device(h)
        import ctypes
        total = ctypes.c_int64()
        assert _lib.lib().tfc_encoder_status(h.ptr, _lib.stream_ptr(), ctypes.byref(total)) != 0
        assert "outgrew its output slab" in _lib.last_error()
        # the ops: fetch_strings / finalize / status repair the handle
        got = tfc.fetch_strings(h)
        assert getattr(h, "retried", False) and [bytes(s) for s in got] == want
        # two calls on one handle (the second appends), through finalize
        h = tfc.create_range_encoder([8], torch.from_numpy(lookup), mode="latency", deferred_errors=True)
        v = torch.from_numpy(value).cuda()
        h = tfc.entropy_encode_channel(tfc.entropy_encode_channel(h, v[:, :12000].contiguous()), v[:, 12000:].contiguous())
        got = tfc.entropy_encode_finalize(h)
        assert getattr(h, "retried", False) and [bytes(s) for s in got] == want
        # several handles behind one launch: only what the data needs is repeated, the strings decode
        hs = tfc.create_range_encoders(2, [8], torch.from_numpy(lookup), mode="latency", deferred_errors=True)
        hs = tfc.entropy_encode_finalize_device_many(tfc.entropy_encode_channel_many(hs, [v, v]))
        for hh in hs:
            assert [bytes(s) for s in tfc.fetch_strings(hh)] == want
    finally:
        del os.environ["TFC_SPECULATIVE_SLAB_DIV"]
    # the same data through a synchronising handle
    h = tfc.create_range_encoder([8], torch.from_numpy(lookup), mode="latency")
    got = tfc.entropy_encode_finalize(tfc.entropy_encode_channel(h, torch.from_numpy(value).cuda()))
    assert [bytes(s) for s in got] == want


def test_outgrown_lane_slab_of_a_model_call_is_coded_again():
    """The lane-per-stream family's deferred slab is 2 bytes per symbol: a latent of mostly far-out values (long escape
    codes) outgrows it.  compress(device_result=True) / compress_many still return the strings of the plain call."""
    torch.manual_seed(1)
    em = tfc.entropy_models.ContinuousBatchedEntropyModel(
        tfc.distributions.NoisyNormal(loc=torch.zeros(8), scale=torch.full((8,), 0.5)), coding_rank=3, compression=True,
        bottleneck_dtype=torch.float32)
    y = (torch.randn(70, 6, 6, 8) * 3000).cuda()           # every symbol an escape code of ~25 bits
    want = em.compress(y)
    h = em.compress(y, device_result=True)
    got = tfc.fetch_strings(h)
    assert [bytes(s) for s in got.reshape(-1)] == [bytes(s) for s in want.reshape(-1)]
    hs = em.compress_many([y, y])
    for hh in hs:
        assert [bytes(s) for s in tfc.fetch_strings(hh).reshape(-1)] == [bytes(s) for s in want.reshape(-1)]
    assert any(getattr(hh, "retried", False) for hh in hs + [h])


def _models():
    torch.manual_seed(0)
    return [
        (tfc.models.BLS2017Model(num_filters=64, compute_dtype=torch.bfloat16).cuda().init_compression(), (96, 128), 12),
        (tfc.models.BMSHJ2018Model(num_filters=64, compute_dtype=torch.bfloat16).cuda().init_compression(), (128, 192), 6),
    ]


@pytest.mark.parametrize("which", [0, 1])
def test_model_device_path_equals_plain_calls(which):
    model, hw, batch = _models()[which]
    x = torch.from_numpy(synthetic.lowpass_images(batch, hw[0], hw[1], seed=9)).cuda()
    plain = model.compress(x)
    want_hat = model.decompress(*plain)
    nstr = model.num_strings
    # inline lane: one stream, nothing read back until fetch
    out = model.compress(x, device_result=True)
    x_hat, oks = model.decompress(*out, defer_sanity=True)
    for k in range(nstr):
        assert [bytes(s) for s in tfc.fetch_strings(out[k])] == [bytes(s) for s in plain[k]]
    assert torch.equal(x_hat, want_hat) and all(bool(ok.cpu().all()) for ok in oks)
    assert out[nstr:] == plain[nstr:]
    # steps in flight on two lanes (one ordinary stream each)
    part = pipeline.StepLanes(2)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        recs = []
        for k in range(4):
            lane = part.lane(k).begin(side)
            o = model.compress(x, device_result=True, lane=lane)
            xh, ok = model.decompress(*o, defer_sanity=True, lane=lane)
            recs.append((o, xh, ok, lane.end_event()))
        for o, xh, ok, end in recs:
            end.synchronize()
            for k in range(nstr):
                assert [bytes(s) for s in tfc.fetch_strings(o[k])] == [bytes(s) for s in plain[k]]
            assert torch.equal(xh, want_hat) and all(bool(f.cpu().all()) for f in ok)
    part.synchronize()


def test_chip_shared_hint_changes_only_the_launch_shape():
    """pipeline.chip_shared(): handles of >= 512 streams are created with two waves per SIMD (tfc_set_chip_shared);
    the strings and the decoded symbols are those of the default layout."""
    _, lookup = _tables()
    lt = torch.from_numpy(lookup)
    value = synthetic.sample_symbols(lookup, 640, 3000, seed=8, escape_fraction=0.004)
    v = torch.from_numpy(value).cuda()
    out = []
    for shared, mode in ((False, "latency"), (True, "latency"), (True, None)):
        # (mode None under the hint: 640 streams go to the lane-per-stream kernels)
        with pipeline.chip_shared(shared):
            h = tfc.entropy_encode_finalize_device(
                tfc.entropy_encode_channel(tfc.create_range_encoder([640], lt, mode=mode, deferred_errors=True), v))
            d, dec = tfc.entropy_decode_channel(tfc.create_range_decoder(h, lt, mode=mode), [3000], torch.int32)
            ok = tfc.entropy_decode_finalize_device(d)
        out.append(([bytes(s) for s in tfc.fetch_strings(h)], dec.cpu(), ok.cpu()))
    for o in out[1:]:
        assert o[0] == out[0][0] and torch.equal(o[1], out[0][1]) and bool(o[2].all())
    assert torch.equal(out[0][1], torch.from_numpy(value))


def test_library_cache_trim():
    """Released coder buffers stay with the library (no hipFreeAsync in steady state); pipeline.empty_cache() hands the
    idle ones back, after which the same calls still work."""
    _, lookup = _tables()
    lt = torch.from_numpy(lookup)
    v = torch.from_numpy(synthetic.sample_symbols(lookup, 24, 5000, seed=3)).cuda()

    def once():
        h = tfc.entropy_encode_finalize_device(tfc.entropy_encode_channel(tfc.create_range_encoder([24], lt, deferred_errors=True), v))
        return [bytes(s) for s in tfc.fetch_strings(h)]

    first = once()
    torch.cuda.synchronize()
    import gc
    gc.collect()
    held = pipeline.cached_bytes()
    assert held > 0
    released = pipeline.empty_cache()
    assert 0 < released <= held and pipeline.cached_bytes() == held - released
    assert once() == first


def test_lane_orders_its_streams():
    """A lane with two streams orders them by events where the stream changes; chip_shared() nests."""
    lane = pipeline.Lane(torch.cuda.Stream(), torch.cuda.Stream())
    a = torch.zeros(1 << 20, device="cuda")
    lane.begin()
    with lane.on("transform"):
        a += 1
    with lane.on("coder"):
        a *= 3
    with lane.on("transform"):
        a -= 1
    lane.join()
    assert float(a.sum()) == 2.0 * (1 << 20)
    from compression_amd import _lib
    with pipeline.chip_shared():
        with pipeline.chip_shared():
            pass
        assert _lib.lib().tfc_set_chip_shared(1) == 1       # the inner context restored the outer's value
    assert _lib.lib().tfc_set_chip_shared(0) == 0


def test_compress_many_equals_batch_by_batch():
    """Several batches through ONE coder launch per direction (lane-per-stream kernels): the strings and the
    reconstructions of compress() / decompress() batch by batch."""
    torch.manual_seed(0)
    model = tfc.models.BLS2017Model(num_filters=64, compute_dtype=torch.bfloat16).cuda().init_compression()
    xs = [torch.from_numpy(synthetic.lowpass_images(6, 64, 96, seed=20 + k)).cuda() for k in range(5)]
    plain = [model.compress(x) for x in xs]
    want = [model.decompress(*p) for p in plain]
    packed = model.compress_many(xs)
    x_hats, ok = model.decompress_many(packed)
    assert bool(ok.cpu().all()) and tuple(ok.shape) == (5, 6)
    for k in range(5):
        assert [bytes(s) for s in tfc.fetch_strings(packed[k][0])] == [bytes(s) for s in plain[k][0]]
        assert packed[k][1:] == plain[k][1:]
        assert torch.equal(x_hats[k], want[k])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_quantise_on_the_lane_kernels(dtype):
    """Throughput-mode handles take bottleneck values through an elementwise quantise pass and the int32 blocks
    (tfc_encoder_encode_quantized[_many], tfc_decoder_decode_dequantized[_many]): same strings and values as the
    wave-per-stream kernels' fused load / store, with a quantisation offset in play."""
    torch.manual_seed(3)
    C = 24
    prior = tfc.NoisyNormal(loc=torch.linspace(-0.7, 0.7, C), scale=torch.linspace(0.3, 7.0, C))
    em = tfc.ContinuousBatchedEntropyModel(prior, coding_rank=3, compression=True, bottleneck_dtype=dtype,
                                           quantization_offset=torch.linspace(-0.4, 0.4, C))
    ys = [(torch.randn(70, 9, 13, C) * torch.linspace(0.3, 7.0, C) * 1.3).to(dtype).cuda() for _ in range(3)]
    tfc.set_default_mode("latency")
    try:
        want = [em.compress(y) for y in ys]
        want_hat = [em.decompress(s, (9, 13)) for s in want]
        tfc.set_default_mode("throughput")
        got = [em.compress(y) for y in ys]
        got_hat = [em.decompress(s, (9, 13)) for s in got]
    finally:
        tfc.set_default_mode("auto")
    handles = em.compress_many(ys)
    many_hat, ok = em.decompress_many(handles, (9, 13))
    assert bool(ok.cpu().all())
    for k in range(3):
        assert [bytes(s) for s in got[k]] == [bytes(s) for s in want[k]]
        assert [bytes(s) for s in tfc.fetch_strings(handles[k])] == [bytes(s) for s in want[k]]
        assert torch.equal(got_hat[k], want_hat[k]) and torch.equal(many_hat[k], want_hat[k])
        assert torch.equal(want_hat[k], em.quantize(ys[k]))
