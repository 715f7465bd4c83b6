"""How often a decoder step's quotient estimate and the exact quotient lie on different sides of a table boundary
(= the hand-scheduled block's verification fails and the block is repeated), on config 2's tables, for
  now    floor(float((D + 1) 2^p) / float(S + 1))        (range_pipe.h, round 5 second pass)
  r04    floor(float((D + 1/2) 2^p) / float(S))          (rounds 3 - 5 first pass; still range_lanes.h's estimate)
against  q* = ceil((D + 1) 2^p / (S + 1)) - 1  (D >= ((S + 1) c) >> 16  <=>  c <= q*), S = span - 1 log-uniform over
[2^16, 2^32), D uniform over [0, S], a random table per sample.  numpy float32 (correctly rounded division where the
GPU has v_rcp_f32 + a multiply: one ulp apart at most).  CPU only:  python tools/r05_estimate_sim.py [samples]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from compression_amd import synthetic          # noqa: E402
from oracle import oracle                      # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
p = 12
rng = np.random.default_rng(1)
port = oracle.port()
pmfs, _ = synthetic.gaussian_pmfs()
tabs = [np.asarray(port.pmf_to_quantized_cdf(pm, p)) for pm in pmfs]
boundary = np.zeros((len(tabs), (1 << p) + 1), bool)
for i, c in enumerate(tabs):
    boundary[i, c[1:-1]] = True
S = np.minimum(np.exp(rng.uniform(np.log(2.0 ** 16), np.log(2.0 ** 32), N)).astype(np.uint64), 2 ** 32 - 1)
D = np.minimum((rng.random(N) * (S + 1)).astype(np.uint64), S)
tab = rng.integers(0, len(tabs), N)
exact = (((D + 1) * (1 << p) + S) // (S + 1) - 1).astype(np.int64)


def estimate(form):
    d, s = D.astype(np.float32), S.astype(np.float32)
    if form == "now":
        num, den = np.float32(d * np.float32(2 ** p) + np.float32(2 ** p)), s + np.float32(1)
    else:
        num, den = np.float32(d * np.float32(2 ** p) + np.float32(2 ** (p - 1))), s
    return np.minimum(np.floor(num * (np.float32(1) / den).astype(np.float32)).astype(np.int64), 1 << p)


for form in ("r04", "now"):
    q = estimate(form)
    lo, hi = np.minimum(q, exact), np.maximum(q, exact)
    fail = np.zeros(N, bool)
    for k in range(1, 4):
        fail |= (hi - lo >= k) & boundary[tab, np.minimum(lo + k, 1 << p)]
    print(f"{form}: estimate != exact quotient for {np.mean(q != exact):.3g} of the samples, across a boundary for {fail.mean():.3g}")
