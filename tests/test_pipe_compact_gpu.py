"""GPU tier: the pipelined decoder on the COMPACT table image (round 6; csrc/range_coder.hip tfc_tables_create "Compact
image", csrc/range_pipe.h TFC_PDEC_STEP_H: every second bound of a row at pair resolution — half the bitmaps — and a
three-way choice among four entries per step; tests/test_pairs_cpu.py holds its arithmetic).  Decoded symbols must equal
the oracle's (cc/kernels/range_coder_kernels.cc:360-471, cc/lib/range_coder.h:193-282) whichever image the chain runs on:
 * the randomised net of tests/test_pipe_fuzz_gpu.py once more with the compact image forced (tfc_set_pipe_format);
 * tables the FULL image of which does not fit a CU — bls2017's shape, 192 rows of 128 symbols: 176 KB — decode on
   dec_chain_kernel (tfc_pipe_counters: launches, nothing left to a fallback), and what the chain gives up on is decoded
   by the wave-per-stream kernel under the job's flag;
 * several chain waves behind one copy of the image (64 batches of BASELINE config 2 as one launch)."""
import ctypes as C

import numpy as np
import pytest
import torch

from compression_amd import synthetic
from test_pipe_fuzz_gpu import CASES, counters, dev, random_lookup, random_values

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tfc():
    import compression_amd
    compression_amd.set_default_mode("throughput")
    yield compression_amd
    compression_amd.set_default_mode("auto")


class pipe_format:
    """with pipe_format(2): ... — the chain's image for the decode launches inside (0 by launch, 1 full, 2 compact)."""

    def __init__(self, fmt, waves=0):
        self.fmt, self.waves = fmt, waves

    def __enter__(self):
        from compression_amd import _lib
        self.prev = _lib.lib().tfc_set_pipe_format(self.fmt, self.waves)
        assert self.prev >= 0

    def __exit__(self, *exc):
        from compression_amd import _lib
        _lib.lib().tfc_set_pipe_format(self.prev, 0)


def decode_oracle_strings(tfc, lookup, strings, elems, index=None):
    arr = np.empty(len(strings), dtype=object)
    for i, x in enumerate(strings):
        arr[i] = x
    hd = tfc.create_range_decoder(arr, torch.from_numpy(lookup))
    if index is None:
        hd, out = tfc.entropy_decode_channel(hd, [elems], torch.int32)
    else:
        hd, out = tfc.entropy_decode_index(hd, dev(index), [elems], torch.int32)
    ok = bool(tfc.entropy_decode_finalize(hd).all())
    return out.cpu().numpy().reshape(len(strings), elems), ok


TOOK = []


@pytest.mark.parametrize("precision,ntab,streams,elems,indexed,esc,seed", [c for c in CASES if c[0] <= 15])
def test_fuzz_net_on_the_compact_image(tfc, port, precision, ntab, streams, elems, indexed, esc, seed):
    rng = np.random.default_rng(seed)
    lookup = random_lookup(port, rng, ntab, precision)
    index = rng.integers(0, ntab, (streams, elems)).astype(np.int32) if indexed else None
    value = random_values(rng, lookup, index, streams, elems, esc)
    strings = port.encode(lookup, value, index=index)[0]
    l0, f0 = counters()
    with pipe_format(2):
        out, ok = decode_oracle_strings(tfc, lookup, strings, elems, index)
    bad = np.argwhere(out != value)
    assert bad.size == 0, (bad[:4], out[tuple(bad[0])], value[tuple(bad[0])])
    assert ok
    l1, f1 = counters()
    TOOK.append((l1 - l0, f1 - f0))


def test_the_compact_chain_took_most_cases():
    clean = [t for t in TOOK if t[0] >= 1 and t[1] == 0]
    assert len(TOOK) >= 60 and len(clean) >= len(TOOK) // 2, (len(TOOK), len(clean))


def wide_tables(port, rows=192, symbols=128, precision=12, seed=0):
    """`rows` tables of exactly `symbols` symbols + the overflow bucket (bls2017's entropy bottleneck has this shape): the
    lane-per-stream image of 192 x 128 is 176 KB — over a CU's LDS."""
    rng = np.random.default_rng(seed)
    cdfs = []
    for r in range(rows):
        x = np.arange(symbols) - (symbols - 1) / 2 + rng.uniform(-3, 3)
        p = np.exp(-0.5 * (x / rng.uniform(2.0, 25.0)) ** 2) + 1e-7
        p = np.concatenate([0.996 * p / p.sum(), [0.004]]).astype(np.float32)
        cdfs.append(port.pmf_to_quantized_cdf(p, precision))
    return synthetic.assemble_lookup(cdfs, precision, overflow=True)


def test_tables_too_wide_for_the_full_image_decode_on_the_pipelined_chain(tfc, port):
    lookup = wide_tables(port)
    streams, elems = 130, 192 * 6 + 5
    value = synthetic.sample_symbols(lookup, streams, elems, seed=3, escape_fraction=0.004)
    strings = port.encode(lookup, value)[0]
    # encode too (the encoder's image is directory + entries: it fitted before), bytes equal
    h = tfc.create_range_encoder([streams], torch.from_numpy(lookup))
    h = tfc.entropy_encode_channel(h, dev(value))
    assert [bytes(x) for x in tfc.entropy_encode_finalize(h).reshape(-1)] == strings
    l0, f0 = counters()
    out, ok = decode_oracle_strings(tfc, lookup, strings, elems)          # format 0: by launch -> compact, nothing else fits
    l1, f1 = counters()
    assert np.array_equal(out, value) and ok
    assert l1 - l0 >= 1 and f1 == f0, "the decode call did not run on dec_chain_kernel"
    # index mode over the same tables
    rng = np.random.default_rng(5)
    index = rng.integers(0, 192, (streams, elems)).astype(np.int32)
    value = random_values(rng, lookup, index, streams, elems, 0.002)
    strings = port.encode(lookup, value, index=index)[0]
    l0, f0 = counters()
    out, ok = decode_oracle_strings(tfc, lookup, strings, elems, index)
    l1, f1 = counters()
    assert np.array_equal(out, value) and ok
    assert l1 - l0 >= 1 and f1 == f0


def test_what_the_chain_gives_up_on_goes_to_the_wave_decoder(tfc, port):
    """One value in five far out: tiles with more escape codes than the launch plans rows for — the chain raises the job's
    flag; with no lane-per-stream image that fits, the wave-per-stream kernel decodes the job under that flag."""
    lookup = wide_tables(port, seed=1)
    streams, elems = 70, 700
    value = synthetic.sample_symbols(lookup, streams, elems, seed=9, escape_fraction=0.3)
    strings = port.encode(lookup, value)[0]
    out, ok = decode_oracle_strings(tfc, lookup, strings, elems)
    assert np.array_equal(out, value) and ok


@pytest.mark.parametrize("waves", [1, 2, 4, 8])
def test_chain_waves_share_one_image(tfc, port, waves):
    """`waves` chain waves per workgroup behind one LDS copy of the compact image (how 64 batches of config 2 are one
    launch: 512 chain waves on 256 CUs)."""
    pmfs, _ = synthetic.gaussian_pmfs(num_tables=48, octave=6.0)
    lookup = synthetic.assemble_lookup([port.pmf_to_quantized_cdf(p, 12) for p in pmfs], 12, overflow=True)
    streams, elems = 64 * 9 + 7, 1000
    value = synthetic.sample_symbols(lookup, streams, elems, seed=waves, escape_fraction=0.01)
    strings = port.encode(lookup, value)[0]
    l0, f0 = counters()
    with pipe_format(2, waves):
        out, ok = decode_oracle_strings(tfc, lookup, strings, elems)
    l1, f1 = counters()
    assert np.array_equal(out, value) and ok
    assert l1 - l0 >= 1 and f1 == f0


def test_full_and_compact_agree_on_config2(tfc, port):
    """BASELINE config 2's tables, one 512-stream batch, both images: the same symbols (and both equal the input)."""
    pmfs, _ = synthetic.gaussian_pmfs()
    lookup = synthetic.assemble_lookup([port.pmf_to_quantized_cdf(p, 12) for p in pmfs], 12, overflow=True)
    streams, elems = 512, 192 * 16
    value = synthetic.sample_symbols(lookup, streams, elems, seed=21, escape_fraction=0.002)
    strings = port.encode(lookup, value, threads=4)[0]
    for fmt in (1, 2):
        with pipe_format(fmt):
            out, ok = decode_oracle_strings(tfc, lookup, strings, elems)
        assert np.array_equal(out, value) and ok, fmt
