"""The pipelined decoder's COMPACT table image (csrc/range_coder.hip tfc_tables_create "Compact image"; csrc/range_pipe.h
TFC_PDEC_STEP_H), restated in numpy and checked exhaustively.  The bitmap marks every SECOND bound of a row (k = o mod 2,
o = symbols mod 2; a one-symbol row: o = -1) at PAIR resolution; for EVERY quotient q the rank i among the marked bounds
whose pair is not behind q's names a window of four entries cdf[k - 1 .. k + 2], k = 2 i + o, and two comparisons of the
quotient with the middle entries give the bounds of the symbol that contains q — the one the full-resolution rank finds,
i.e. the symbol the coder's search finds for an offset whose exact quotient is q (tests/test_estimate_cpu.py) — with the
raw entry 2 i + t0 + t1 = symbol - (o - 1).  The row's end (2^16, stored as 0) must never meet a comparison.
Rows: the oracle's tables for BASELINE config 2's Gaussians, and random strictly increasing rows dense in width-1 symbols
(runs of consecutive bounds: what one bit per pair cannot tell apart), precisions 1 ... 15."""
import numpy as np
import pytest

from compression_amd import synthetic
from oracle import oracle


def compact_row(cdf, p):
    """-> (entries [2 pads + n + 1 (+ 2 behind)], index of the window of rank 0, marks bool [pairs], o)."""
    sh = 16 - p
    n = len(cdf) - 1
    o = -1 if n == 1 else n & 1
    entries = np.concatenate([[0, 0], (np.asarray(cdf, np.int64) << sh) & 0xFFFF, [0, 0]])
    window0 = 2 + o - 1
    marks = np.zeros(max(1, 1 << (p - 1)) + 1, bool)
    for k in range(o + 2, n, 2):                 # the first marked bound, k = o, is left out: the rank is its successors' count
        assert not marks[cdf[k] >> 1]            # two marked bounds never share a pair
        marks[cdf[k] >> 1] = True
    return entries, window0, marks, o


def check_row(cdf, p):
    cdf = np.asarray(cdf, np.int64)
    n = len(cdf) - 1
    assert cdf[0] == 0 and cdf[-1] == 1 << p and np.all(np.diff(cdf) > 0)
    sh = 16 - p
    entries, window0, marks, o = compact_row(cdf, p)
    q = np.arange((1 << p) + 1, dtype=np.int64)                    # (2^p: the estimate of an offset at the span's very top)
    want = np.minimum(np.searchsorted(cdf, q, side="right") - 1, n - 1)
    i = np.cumsum(marks)[q >> 1]
    base = window0 + 2 * i
    e0, e1, e2, e3 = (entries[base + d] for d in range(4))
    Q = q << sh
    t0 = (Q >= e1).astype(np.int64)
    t1 = (Q >= e2).astype(np.int64)
    assert np.all(t1 <= t0)
    lower = np.where(t1 == 1, e2, np.where(t0 == 1, e1, e0))
    upper = np.where(t1 == 1, e3, np.where(t0 == 1, e2, e1))
    assert np.array_equal(lower, (cdf[want] << sh) & 0xFFFF)
    assert np.array_equal(upper, (cdf[want + 1] << sh) & 0xFFFF)
    assert np.array_equal(2 * i + t0 + t1 + (o - 1), want)
    assert (2 * i + t0 + t1).max() < 0x8000                        # a raw entry: bit 15 is the bit rows'
    # the end entry (2^16 stored as 0) is compared nowhere: the compared entries are bounds below 2^p, or pads
    end_index = 2 + n
    assert np.all(base + 1 != end_index) and np.all(base + 2 != end_index)
    # windows start at even entry distances from an aligned first one (ds_read2_b32)
    assert np.all(base - window0 >= 0)


def test_config2_tables():
    port = oracle.port()
    pmfs, _ = synthetic.gaussian_pmfs()
    for pm in pmfs[::5]:
        check_row(port.pmf_to_quantized_cdf(pm, 12), 12)


@pytest.mark.parametrize("p", list(range(1, 16)))
def test_random_rows_dense_in_width_one_symbols(p):
    rng = np.random.default_rng(p)
    top = 1 << p
    for trial in range(14):
        if trial == 0:
            inner = np.arange(1, top)                              # every quotient value a bound
        elif trial == 1:
            inner = np.array([], np.int64)                         # one symbol
        elif trial == 2:
            inner = np.arange(1, top)[:1]                          # two symbols, the first of width one
        else:
            keep = rng.random(top - 1) < rng.choice([0.02, 0.2, 0.6])
            runs = np.zeros(top - 1, bool)
            for _ in range(rng.integers(0, 6)):
                a = rng.integers(0, top - 1)
                runs[a:a + rng.integers(1, 40)] = True
            inner = np.arange(1, top)[keep | runs]
            if trial % 2:
                inner = inner[:len(inner) - 1]                     # both parities of the symbol count
        inner = inner[:32766]                                      # (rows of at most 32 767 symbols take these kernels)
        check_row(np.concatenate([[0], inner, [top]]).astype(np.int64), p)


def test_binary_row_entries():
    """The built-in row of an escape code's bits {0, 1/2, 1}, ranks carried by 0x4000: the raw entry is 0x8001 + bit."""
    for p in range(1, 16):
        cdf = np.array([0, 1 << (p - 1), 1 << p])
        entries, window0, marks, o = compact_row(cdf, p)
        assert o == 0 and not marks.any()
        sh = 16 - p
        for q in range(1 << p):
            e1, e2 = entries[window0 + 1], entries[window0 + 2]
            t0, t1 = int((q << sh) >= e1), int((q << sh) >= e2)
            raw = (2 * 0x4000 + t0 + t1) & 0xFFFF
            assert raw == 0x8001 + int(q >= 1 << (p - 1)) and (raw >> 1) & 1 == int(q >= 1 << (p - 1))
