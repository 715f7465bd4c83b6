"""GPU tier: the device-side table build (SURVEY §8 a2 — continuous_base.py:217-296): tfc_build_tables (all rows of a
model in one launch) against the oracle's PmfToQuantizedCdf on the same float32 masses, and tfc_deep_factorized_tails
(helpers.estimate_tails as one kernel) against the tensor-op iteration it replaces."""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def overflow_like_the_kernel(p):
    """max(1 - sum(p), 0) in float32 in the order csrc/pmf_to_cdf.hip states: lane l adds elements l, l + 64, ... in
    turn, then the 64 partial sums are combined by the xor butterfly 32, 16, ..., 1."""
    part = np.zeros(64, np.float32)
    for i, x in enumerate(np.asarray(p, np.float32)):
        part[i % 64] = np.float32(part[i % 64] + x)
    lanes = np.arange(64)
    for off in (32, 16, 8, 4, 2, 1):
        part = (part + part[lanes ^ off]).astype(np.float32)
    return np.float32(max(np.float32(1.0) - part[0], np.float32(0.0)))


def build(pmf_rows, precision):
    """tfc_build_tables on ragged rows -> list of [-precision, cdf...] int32 arrays."""
    from compression_amd import _lib
    rows = len(pmf_rows)
    lengths = np.array([len(p) for p in pmf_rows], np.int32)
    stride = int(lengths.max())
    pmf = np.zeros((rows, stride), np.float32)
    for r, p in enumerate(pmf_rows):
        pmf[r, :len(p)] = p
    ends = np.cumsum(lengths.astype(np.int64) + 3)
    offsets = ends - (lengths + 3)
    d_pmf, d_len, d_off = (torch.from_numpy(a).cuda() for a in (pmf, lengths, offsets))
    out = torch.full((int(ends[-1]),), -12345, dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib().tfc_build_tables(d_pmf.data_ptr(), rows, stride, d_len.data_ptr(), d_off.data_ptr(), stride,
                                           precision, out.data_ptr(), _lib.stream_ptr()))
    out = out.cpu().numpy()
    return [out[offsets[r]:ends[r]] for r in range(rows)]


@pytest.mark.parametrize("precision", [12, 16, 7])
def test_build_tables_equals_the_oracle_row_by_row(port, precision):
    rng = np.random.default_rng(precision)
    rows = []
    for n in (1, 2, 3, 17, 63, 64, 65, 130, 200, 500):
        if n + 1 > (1 << precision):
            # more symbols than quanta: the reference aborts there (CHECK at pmf_to_cdf_kernels.cc:110, every symbol
            # must keep a frequency >= 1), so there is nothing to compare with
            continue
        for kind in range(3):
            if kind == 0:        # a discretised Gaussian with mass left over (the usual row)
                x = np.arange(n) - (n - 1) / 2
                p = np.exp(-0.5 * (x / max(n / 6, 0.3)) ** 2)
                p = 0.996 * p / p.sum()
            elif kind == 1:      # exactly symmetric, ties everywhere
                p = np.ones(n) * (0.9 / n)
            else:                # random, sums past 1 (overflow clamps to 0)
                p = rng.random(n)
                p = 1.02 * p / p.sum()
            rows.append(p.astype(np.float32))
    got = build(rows, precision)
    for p, g in zip(rows, got):
        full = np.concatenate([p, [overflow_like_the_kernel(p)]]).astype(np.float32)
        want = np.asarray(port.pmf_to_quantized_cdf(full, precision), np.int32)
        assert g[0] == -precision
        assert np.array_equal(g[1:], want), (len(p), precision)
        assert g[1] == 0 and g[-1] == 1 << precision and np.all(np.diff(g[1:]) > 0)


def test_model_tables_equal_the_oracle_and_build_in_milliseconds(port):
    """bls2017's entropy bottleneck (192 deep factorized channels) and bmshj2018's 64 scale tables: every row of the
    stored table equals the oracle's PmfToQuantizedCdf of the model's own float32 masses; building them is a handful of
    launches (round 4: 18 000)."""
    import compression_amd as tfc
    torch.manual_seed(0)
    for make in (lambda: tfc.models.BLS2017Model(num_filters=192).cuda(), lambda: tfc.models.BMSHJ2018Model(num_filters=192).cuda()):
        model = make()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.init_compression()
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
        t0 = time.perf_counter()
        model.init_compression()
        torch.cuda.synchronize()
        again = time.perf_counter() - t0
        print(f"{type(model).__name__}.init_compression(): {1e3 * first:.1f} ms the first time, {1e3 * again:.1f} ms again")
        assert again < 0.25
        from compression_amd import synthetic
        for em in {id(m): m for m in (model.entropy_model, getattr(model, "side_entropy_model", None)) if m is not None}.values():
            prior, precision = em.prior, em.range_coder_precision
            table, minima = em.cdf.cpu().numpy(), em.cdf_offset.cpu().numpy()
            rows = synthetic.lookup_rows(table)
            lengths = np.array([len(c) - 2 for _, c in rows])           # masses per row (the cdf has the overflow's + 1 more)
            assert len(rows) == len(minima) and all(sp == -precision for sp, _ in rows)
            qoff = getattr(em, "quantization_offset", None)
            with torch.no_grad():
                start = torch.from_numpy(minima).to(prior.dtype)
                if qoff is not None:
                    start = start + qoff.reshape(-1).cpu().to(prior.dtype)
                batch = tuple(prior.batch_shape)
                samples = (torch.arange(int(lengths.max()), dtype=prior.dtype).reshape((-1,) + len(batch) * (1,))
                           + start.reshape(batch)).cuda()
                pmf = prior.prob(samples).reshape(int(lengths.max()), -1).t().float().cpu().numpy()
            for r, (_, c) in enumerate(rows):
                p = pmf[r, :lengths[r]]
                full = np.concatenate([p, [overflow_like_the_kernel(p)]]).astype(np.float32)
                want = np.asarray(port.pmf_to_quantized_cdf(full, precision), np.int32)
                assert np.array_equal(np.asarray(c, np.int32), want), (type(em).__name__, r)


@pytest.mark.parametrize("num_filters", [(3, 3), (3, 3, 3), (5, 5)])
def test_device_tails_equal_the_tensor_op_iteration(num_filters):
    """tfc_deep_factorized_tails against helpers.estimate_tails on the same prior: the same solution (the iteration's
    tolerance is 1e-8 on the logits; the two differ by float32 rounding of the derivative), and the same table support."""
    from compression_amd import distributions
    from compression_amd.distributions import helpers
    torch.manual_seed(7)
    prior = distributions.DeepFactorized(batch_shape=(96,), num_filters=num_filters)
    with torch.no_grad():
        for m in prior.matrices:
            m.add_(0.3 * torch.randn_like(m))
        for f in prior.factors:
            f.add_(0.5 * torch.randn_like(f))
    prior = prior.cuda()
    import math
    for target in (0.0, math.log(2.0 ** -9 / (1 - 2.0 ** -9)), -math.log(2.0 ** -9 / (1 - 2.0 ** -9))):
        got = prior._solve_device([target])
        assert got is not None
        want = helpers.estimate_tails(prior._logits_cumulative, target, prior.batch_shape, prior.dtype, "cuda")
        with torch.no_grad():
            res = (prior._logits_cumulative(got[0]) - target).abs().max().item()
            res_want = (prior._logits_cumulative(want) - target).abs().max().item()
        assert res <= max(2e-6, 4 * res_want), (res, res_want)
        assert torch.allclose(got[0], want, atol=2e-3, rtol=1e-4)


def test_invalid_masses_are_reported_like_the_op():
    import compression_amd as tfc

    class Broken(tfc.distributions.NoisyNormal):
        def _prob(self, y):
            return super()._prob(y) * float("nan")
    with pytest.raises(ValueError, match="non-finite or negative element"):
        tfc.entropy_models.ContinuousBatchedEntropyModel(Broken(loc=torch.zeros(4).cuda(), scale=torch.ones(4).cuda()),
                                                        coding_rank=1, compression=True)


def test_float64_prior_sums_its_overflow_in_float64(port):
    """continuous_base.py:277-279: max(1 - reduce_sum(p), 0) in the PRIOR's dtype, the cast to float32 afterwards — a
    float64 prior's rows equal the oracle's PmfToQuantizedCdf of float32(p) with float32(max(1 - sum64(p), 0)) appended
    (tfc_build_tables_overflow), which differs from the float32 sum's table where the two overflow masses round apart."""
    import compression_amd as tfc
    from compression_amd import synthetic
    scale = torch.linspace(0.3, 40.0, 48, dtype=torch.float64).cuda()
    prior = tfc.distributions.NoisyNormal(loc=torch.zeros_like(scale), scale=scale, dtype=torch.float64)
    assert prior.dtype == torch.float64
    em = tfc.entropy_models.ContinuousBatchedEntropyModel(prior, coding_rank=1, compression=True)
    precision = em.range_coder_precision
    table, minima = em.cdf.cpu().numpy(), em.cdf_offset.cpu().numpy()
    rows = synthetic.lookup_rows(table)
    assert len(rows) == 48
    lengths = np.array([len(c) - 2 for _, c in rows])
    qoff = getattr(em, "quantization_offset", None)
    with torch.no_grad():
        start = torch.from_numpy(minima).to(torch.float64)
        if qoff is not None:
            start = start + qoff.reshape(-1).cpu().to(torch.float64)
        samples = (torch.arange(int(lengths.max()), dtype=torch.float64).reshape(-1, 1) + start.reshape(1, -1)).cuda()
        pmf = prior.prob(samples).t().cpu().numpy()
    assert pmf.dtype == np.float64
    differ = 0
    for r, (sp, c) in enumerate(rows):
        p64 = pmf[r, :lengths[r]]
        overflow = np.float32(max(1.0 - p64.sum(dtype=np.float64), 0.0))
        differ += int(overflow != overflow_like_the_kernel(p64.astype(np.float32)))
        full = np.concatenate([p64.astype(np.float32), [overflow]]).astype(np.float32)
        want = np.asarray(port.pmf_to_quantized_cdf(full, precision), np.int32)
        assert sp == -precision and np.array_equal(np.asarray(c, np.int32), want), r
    print(f"rows whose float64 and float32 overflow masses differ as float32: {differ} of {len(rows)}")
