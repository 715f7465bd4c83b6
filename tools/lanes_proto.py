"""numpy model of the lane-per-stream decoder step (csrc/range_lanes.h): every array element is one
lane = one code stream.  Used to check the arithmetic of the symbol-first search (float quotient
estimate -> rank in the row's boundary bitmap -> exact verification) against the oracle and to count
how often the verification has to correct the estimate.  Run: python tools/lanes_proto.py"""
import sys
import os
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from compression_amd import synthetic  # noqa: E402


def build_image(lookup):
    rows = synthetic.lookup_rows(lookup)
    img = []
    for sp, cdf in rows:
        p = abs(sp)
        cdf = np.asarray(cdf, np.int64)
        nsym = len(cdf) - 1
        words = max(1, (1 << p) // 64)
        bits = np.zeros(words, np.uint64)
        for k in range(nsym):
            bits[cdf[k] >> 6] |= np.uint64(1) << np.uint64(cdf[k] & 63)
        cum = np.zeros(words, np.int64)
        cum[1:] = np.cumsum([bin(int(b)).count("1") for b in bits])[:-1]
        img.append(dict(p=p, esc=sp < 0, nsym=nsym, cdf16=(cdf << (16 - p)), bits=bits, cum=cum))
    return img


def popcount64(x):
    x = x.astype(np.uint64)
    c = np.zeros(x.shape, np.int64)
    for i in range(64):
        c += ((x >> np.uint64(i)) & np.uint64(1)).astype(np.int64)
    return c


def decode(lookup, strings, elems, add_one=True, bias=0.5):
    """Channel-mode decode of len(strings) streams in lockstep (no escapes supported in the model)."""
    img = build_image(lookup)
    n = len(strings)
    maxlen = max(len(s) for s in strings) + 8
    buf = np.zeros((n, maxlen), np.uint8)
    for i, s in enumerate(strings):
        buf[i, :len(s)] = np.frombuffer(s, np.uint8)
    pos = np.full(n, 4, np.int64)
    window = (buf[:, 0].astype(np.uint64) << 24 | buf[:, 1].astype(np.uint64) << 16 |
              buf[:, 2].astype(np.uint64) << 8 | buf[:, 3].astype(np.uint64))
    D = window.astype(np.uint64)          # base = 0
    s1 = np.full(n, 0xFFFFFFFF, np.uint64)
    out = np.zeros((n, elems), np.int32)
    fixes = 0
    ar = np.arange(n)
    for j in range(elems):
        r = img[j % len(img)]
        p = r["p"]
        fD = D.astype(np.float32) + np.float32(bias)
        fS = s1.astype(np.float32) + (np.float32(1.0) if add_one else np.float32(0.0))
        q = (fD * (np.float32(1.0) / fS)) * np.float32(1 << p)
        cp = np.minimum(q.astype(np.int64), (1 << p) - 1)
        w = cp >> 6
        bit = cp & 63
        mask = (~np.uint64(0)) >> (np.uint64(63) - bit.astype(np.uint64))
        rank = r["cum"][w] + popcount64(r["bits"][w] & mask)
        s = rank - 1
        span = s1 + np.uint64(1)
        for _ in range(4):
            lo = r["cdf16"][s].astype(np.uint64)
            hi = r["cdf16"][s + 1].astype(np.uint64)
            A = (span * lo) >> np.uint64(16)
            B = (span * hi) >> np.uint64(16)
            bad_lo = D < A
            bad_hi = D >= B
            if not (bad_lo.any() or bad_hi.any()):
                break
            fixes += int(bad_lo.sum() + bad_hi.sum())
            s = np.clip(s - bad_lo + bad_hi, 0, r["nsym"] - 1)
        Dn = D - A
        sn = B - A - np.uint64(1)
        ren = sn < 65536
        dig = buf[ar, pos].astype(np.uint64) << 8 | buf[ar, pos + 1].astype(np.uint64)
        D = np.where(ren, (Dn << np.uint64(16)) | dig, Dn) & np.uint64(0xFFFFFFFF)
        s1 = np.where(ren, (sn << np.uint64(16)) | np.uint64(0xFFFF), sn)
        pos = pos + 2 * ren
        out[:, j] = s
    return out, fixes


def build_records(lookup):
    """One record per 64-position word of a row, everything the step needs from ONE LDS read:
    the boundary bits of the word, the number of boundaries before it (cum), the last boundary at or before
    the word's first position (lo_in) and the first boundary after the word (hi_out).  From the record of the
    word that holds the estimate cp:
        masked = bits & (all ones up to bit cp)          lo = masked ? position of its top bit : lo_in
        above  = bits & ~(all ones up to bit cp)         hi = above  ? position of its low bit : hi_out
        s      = cum + popcount(masked) - 1
    i.e. the symbol AND its exact interval without the dependent read of cdf[s], cdf[s + 1]."""
    rows = synthetic.lookup_rows(lookup)
    rec = []
    for sp, cdf in rows:
        p = abs(sp)
        cdf = np.asarray(cdf, np.int64)
        nsym = len(cdf) - 1
        words = max(1, (1 << p) // 64)
        bits = np.zeros(words, np.uint64)
        for k in range(nsym):
            bits[cdf[k] >> 6] |= np.uint64(1) << np.uint64(cdf[k] & 63)
        starts = np.arange(words, dtype=np.int64) * 64
        cum = np.searchsorted(cdf[:nsym], starts, side="left")              # boundaries strictly before the word
        lo_in = cdf[np.maximum(np.searchsorted(cdf[:nsym], starts, side="right") - 1, 0)]
        hi_out = cdf[np.searchsorted(cdf, starts + 63, side="right").clip(max=nsym)]
        rec.append(dict(p=p, nsym=nsym, bits=bits, cum=cum, lo_in=lo_in, hi_out=hi_out, cdf=cdf))
    return rec


def _top_bit(x):
    """Position of the highest set bit of each uint64 (x != 0)."""
    pos = np.zeros(x.shape, np.int64)
    v = x.copy()
    for sh in (32, 16, 8, 4, 2, 1):
        big = (v >> np.uint64(sh)) != 0
        pos += sh * big
        v = np.where(big, v >> np.uint64(sh), v)
    return pos


def _low_bit(x):
    return _top_bit(x & (~x + np.uint64(1)))


def decode_onetrip(lookup, strings, elems):
    """The same lockstep decode with lo / hi taken from the word record (build_records) instead of the
    row's cdf entries; returns (symbols, corrections, share of steps whose lo and hi both came out of the
    bitmap word itself)."""
    rec = build_records(lookup)
    n = len(strings)
    maxlen = max(len(s) for s in strings) + 8
    buf = np.zeros((n, maxlen), np.uint8)
    for i, s in enumerate(strings):
        buf[i, :len(s)] = np.frombuffer(s, np.uint8)
    pos = np.full(n, 4, np.int64)
    D = (buf[:, 0].astype(np.uint64) << 24 | buf[:, 1].astype(np.uint64) << 16 |
         buf[:, 2].astype(np.uint64) << 8 | buf[:, 3].astype(np.uint64))
    s1 = np.full(n, 0xFFFFFFFF, np.uint64)
    out = np.zeros((n, elems), np.int32)
    fixes = in_word = 0
    ar = np.arange(n)
    for j in range(elems):
        r = rec[j % len(rec)]
        p = r["p"]
        q = ((D.astype(np.float32) + np.float32(0.5)) * (np.float32(1.0) / (s1.astype(np.float32) + np.float32(1.0)))
             ) * np.float32(1 << p)
        cp = np.minimum(q.astype(np.int64), (1 << p) - 1)
        span = s1 + np.uint64(1)
        for attempt in range(6):
            w, bit = cp >> 6, (cp & 63).astype(np.uint64)
            word = r["bits"][w]
            upto = (~np.uint64(0)) >> (np.uint64(63) - bit)
            masked, above = word & upto, word & ~upto
            lo = np.where(masked != 0, (w << 6) + _top_bit(np.where(masked != 0, masked, np.uint64(1))), r["lo_in"][w])
            hi = np.where(above != 0, (w << 6) + _low_bit(np.where(above != 0, above, np.uint64(1))), r["hi_out"][w])
            s = r["cum"][w] + popcount64(masked) - 1
            A = (span * (lo.astype(np.uint64) << np.uint64(16 - p))) >> np.uint64(16)
            B = (span * (hi.astype(np.uint64) << np.uint64(16 - p))) >> np.uint64(16)
            bad_lo, bad_hi = D < A, D >= B
            if attempt == 0:
                in_word += int(((masked != 0) & (above != 0)).sum())
            if not (bad_lo.any() or bad_hi.any()):
                break
            fixes += int(bad_lo.sum() + bad_hi.sum())
            # a rejected estimate moves to the neighbouring symbol: just below lo / at hi
            cp = np.where(bad_lo, np.maximum(lo - 1, 0), np.where(bad_hi, np.minimum(hi, (1 << p) - 1), cp))
        assert np.array_equal(lo, r["cdf"][s]) and np.array_equal(hi, r["cdf"][s + 1])
        Dn = D - A
        sn = B - A - np.uint64(1)
        ren = sn < 65536
        dig = buf[ar, pos].astype(np.uint64) << 8 | buf[ar, pos + 1].astype(np.uint64)
        D = np.where(ren, (Dn << np.uint64(16)) | dig, Dn) & np.uint64(0xFFFFFFFF)
        s1 = np.where(ren, (sn << np.uint64(16)) | np.uint64(0xFFFF), sn)
        pos = pos + 2 * ren
        out[:, j] = s
    return out, fixes, in_word / (n * elems)


def main():
    from oracle import oracle
    port = oracle.port()
    pmfs, _ = synthetic.gaussian_pmfs(192)
    cdfs = [port.pmf_to_quantized_cdf(p[None, :], 12)[0] for p in pmfs]
    lookup = synthetic.assemble_lookup(cdfs, 12)
    streams, elems = 64, 4096
    sym = synthetic.sample_symbols(lookup, streams, elems, seed=0)
    strings, _, _ = port.encode(lookup, sym)
    total = sum(len(c) for c in cdfs)
    print("table entries", total, "rows", len(cdfs), "max row", max(len(c) for c in cdfs))
    for add_one in (True, False):
        for bias in (0.0, 0.5, 1.0):
            got, fixes = decode(lookup, strings, elems, add_one, bias)
            print(f"add_one={add_one} bias={bias}: exact={np.array_equal(got, sym)} "
                  f"corrections={fixes} of {streams * elems} ({fixes / (streams * elems):.2e})")
    got, fixes, share = decode_onetrip(lookup, strings, elems)
    rec = build_records(lookup)
    words = sum(len(r["bits"]) for r in rec)
    print(f"one-trip records: exact={np.array_equal(got, sym)} corrections={fixes} "
          f"({fixes / (streams * elems):.2e}); lo and hi both inside the word in {100 * share:.1f} % of the steps; "
          f"{words} words: {words * 16 / 1024:.0f} KB at 16 B per word, {words * 12 / 1024:.0f} KB at 12 B")


if __name__ == "__main__":
    main()
