"""CPU tier: host logic of the entropy models and distributions — ports of the
reference tests that do not need the coder (continuous_batched_test.py:26-101,
helpers_test.py, deep_factorized_test.py, math_ops_test.py at reduced size)."""
import math

import numpy as np
import pytest
import torch

import compression_amd as tfc
from compression_amd.distributions import helpers


def test_can_instantiate():
    noisy = tfc.NoisyNormal(loc=0.0, scale=1.0)
    em = tfc.ContinuousBatchedEntropyModel(noisy, 1)
    assert em.prior is noisy and em.coding_rank == 1 and em.tail_mass == 2 ** -8
    assert em.bottleneck_dtype == torch.float32 and em.prior.dtype == torch.float32


def test_requires_coding_rank_bigger_than_prior_batch_rank():
    noisy = tfc.NoisyLogistic(loc=0.0, scale=torch.tensor([[1.0], [2.0]]))
    for rank in (0, 1):
        with pytest.raises(ValueError):
            tfc.ContinuousBatchedEntropyModel(noisy, rank)
    tfc.ContinuousBatchedEntropyModel(noisy, 2)
    tfc.ContinuousBatchedEntropyModel(noisy, 3)


def test_argument_validation():
    noisy = tfc.NoisyNormal(loc=0.0, scale=1.0)
    with pytest.raises(ValueError, match="Either `prior` or `prior_shape`"):
        tfc.ContinuousBatchedEntropyModel(coding_rank=1)
    with pytest.raises(ValueError, match="CDFs can't be provided"):
        tfc.ContinuousBatchedEntropyModel(prior_shape=(), coding_rank=1, cdf_shapes=(3, 1),
                                          quantization_offset=False)
    with pytest.raises(ValueError, match="tail_mass"):
        tfc.ContinuousBatchedEntropyModel(noisy, 1, tail_mass=1.5)


def test_quantizes_to_integers_modulo_offset():
    noisy = tfc.NoisyNormal(loc=0.25, scale=10.0)
    em = tfc.ContinuousBatchedEntropyModel(noisy, 1)
    x = torch.arange(-20.0, 20.0) + 0.25
    xp = x + (torch.rand(x.shape) * 0.98 - 0.49)
    assert torch.equal(em.quantize(xp), x)


def test_gradients_are_straight_through():
    em = tfc.ContinuousBatchedEntropyModel(tfc.NoisyNormal(loc=0.0, scale=1.0), 1)
    xp = (torch.arange(-20.0, 20.0) + torch.rand(40) * 0.98 - 0.49).requires_grad_()
    em.quantize(xp).sum().backward()
    assert torch.equal(xp.grad, torch.ones_like(xp))


def test_default_kwargs_throw_error_on_compression():
    em = tfc.ContinuousBatchedEntropyModel(tfc.NoisyNormal(loc=0.25, scale=10.0), 1)
    with pytest.raises(RuntimeError):
        em.compress(torch.zeros(10))
    with pytest.raises(RuntimeError):
        em.decompress(np.array([b""] * 10, dtype=object), [10])


def test_rate_bounds_without_coder():
    # training bits >= eval bits (asymptotically); tight for wide priors
    torch.manual_seed(0)
    for scale, tight in ((0.25, False), (64.0, True)):
        prior = tfc.NoisyNormal(loc=0.5, scale=scale)
        em = tfc.ContinuousBatchedEntropyModel(prior, coding_rank=1)
        x = 0.5 + scale * torch.randn(200000)
        _, be = em(x, training=False)
        _, bt = em(x, training=True)
        assert bt > 0.999 * be
        if tight:
            assert abs(float(bt) - float(be)) <= 1e-3 * float(be)


def test_laplace_tail_mass_and_expected_grads():
    noisy = tfc.NoisyNormal(loc=0.0, scale=1.0)
    em = tfc.ContinuousBatchedEntropyModel(noisy, 1, laplace_tail_mass=1e-3, expected_grads=True)
    x = torch.randn(4, 50, requires_grad=True)
    _, bits = em(x, training=True)
    bits.sum().backward()
    assert torch.isfinite(x.grad).all() and (bits > 0).all()


def test_bounds_gradients():
    x = torch.tensor([-1.0, 0.5, 2.0], requires_grad=True)
    y = tfc.lower_bound(x, 0.0)
    y.backward(torch.tensor([1.0, 1.0, 1.0]))       # positive grad pushes away from the bound
    assert torch.equal(y.detach(), torch.tensor([0.0, 0.5, 2.0]))
    assert torch.equal(x.grad, torch.tensor([0.0, 1.0, 1.0]))
    x.grad = None
    tfc.lower_bound(x, 0.0).backward(torch.tensor([-1.0, -1.0, -1.0]))   # towards the bound: passes
    assert torch.equal(x.grad, torch.tensor([-1.0, -1.0, -1.0]))
    x.grad = None
    tfc.upper_bound(x, 1.0, gradient="disconnected").backward(torch.ones(3))
    assert torch.equal(x.grad, torch.tensor([1.0, 1.0, 0.0]))
    with pytest.raises(ValueError):
        tfc.lower_bound(x, 0.0, gradient="bogus")


def test_same_padding_for_kernel():
    assert tfc.same_padding_for_kernel((5, 5), True) == [(2, 2), (2, 2)]
    assert tfc.same_padding_for_kernel((4,), True) == [(2, 1)]
    assert tfc.same_padding_for_kernel((4,), False) == [(1, 2)]
    assert tfc.same_padding_for_kernel((9, 9), False, (4, 4)) == [(1, 1), (1, 1)]
    assert tfc.same_padding_for_kernel((5,), False, (2,)) == [(1, 1)]


def test_estimate_tails_and_helpers():
    d = tfc.Logistic(loc=torch.tensor([0.0, 3.0]), scale=torch.tensor([1.0, 2.0]))

    class NoQuantile(type(d)):
        def _quantile(self, q):
            raise NotImplementedError

    nq = NoQuantile(loc=d.loc, scale=d.scale)
    got = helpers.lower_tail(nq, 2 ** -8)
    want = helpers.lower_tail(d, 2 ** -8)
    assert torch.allclose(got, want, atol=1e-2)
    got = helpers.upper_tail(nq, 2 ** -8)
    assert torch.allclose(got, helpers.upper_tail(d, 2 ** -8), atol=1e-2)
    off = helpers.quantization_offset(tfc.NoisyNormal(loc=torch.tensor([0.25, 3.75]), scale=1.0))
    assert torch.allclose(off, torch.tensor([0.25, -0.25]))


def test_deep_factorized_is_a_density():
    torch.manual_seed(1)
    df = tfc.NoisyDeepFactorized(batch_shape=(3,))
    x = torch.arange(-200.0, 201.0)[:, None]
    p = df.prob(x)
    assert (p >= 0).all() and torch.allclose(p.sum(0), torch.ones(3), atol=1e-3)
    assert torch.allclose(df.log_prob(x).exp(), p, atol=1e-6)
    lt, ut = helpers.lower_tail(df, 2 ** -8), helpers.upper_tail(df, 2 ** -8)
    assert torch.allclose(df.base.cdf(lt), torch.full((3,), 2 ** -9), atol=2e-4)
    assert torch.allclose(df.base.survival_function(ut), torch.full((3,), 2 ** -9), atol=2e-4)
    # state dict keys follow the reference's variable names
    keys = set(df.state_dict().keys())
    assert {"base.matrices.0", "base.biases.0", "base.factors.0"} <= keys


def test_noisy_normal_prob_matches_cdf_difference():
    d = tfc.NoisyNormal(loc=0.3, scale=torch.tensor([0.5, 4.0]))
    x = torch.linspace(-10, 10, 41)[:, None]
    want = d.base.cdf(x + 0.5) - d.base.cdf(x - 0.5)
    assert torch.allclose(d.prob(x), want, atol=1e-6)
    assert torch.allclose(d.log_prob(x).exp(), want, atol=1e-6)


def test_noisy_laplace_log_prob_keeps_its_digits_in_the_tails():
    """The Laplace log cdf / log survival function are evaluated without cancellation (as tfp's, which the
    reference uses): the unit-interval log mass equals its closed form log(sinh(.5)) - |v| at any distance,
    in float32 — this is the branch the Laplace-mixture tail (continuous_base.py:298-334) takes far out."""
    d = tfc.NoisyLaplace(loc=0.0, scale=1.0)
    v = torch.tensor([-80.0, -30.0, -17.0, -3.0, -0.75, 3.0, 17.0, 30.0, 80.0], requires_grad=True)
    lp = d.log_prob(v)
    want = math.log(math.sinh(0.5)) - v.detach().abs()
    assert torch.allclose(lp, want, rtol=1e-6, atol=1e-6)
    lp.sum().backward()
    assert torch.equal(v.grad, -torch.sign(v.detach()))
    near = torch.linspace(-0.45, 0.45, 7)
    assert torch.allclose(d.log_prob(near), torch.log(1 - math.exp(-0.5) * torch.cosh(near)), atol=1e-6)
    # a scaled, shifted one against the float64 cdf difference
    e = tfc.NoisyLaplace(loc=0.7, scale=torch.tensor([0.3, 2.5]))
    x = torch.linspace(-12, 12, 25)[:, None]
    e64 = tfc.NoisyLaplace(loc=0.7, scale=torch.tensor([0.3, 2.5], dtype=torch.float64), dtype=torch.float64)
    want = torch.log(e64.base.cdf(x.double() + 0.5) - e64.base.cdf(x.double() - 0.5))
    ok = torch.isfinite(want) & (want > -30)
    assert torch.allclose(e.log_prob(x).double()[ok], want[ok], rtol=1e-5, atol=1e-5)


def test_indexed_model_host_logic():
    em = tfc.LocationScaleIndexedEntropyModel(
        tfc.NoisyNormal, num_scales=8, scale_fn=lambda i: torch.exp(math.log(0.5) + 0.4 * i), coding_rank=1)
    idx = torch.tensor([[-3.0, 0.0, 7.0, 11.0]])
    assert torch.equal(em._normalize_indexes(idx), torch.tensor([[0.0, 0.0, 7.0, 7.0]]))
    x = torch.randn(1, 4)
    y, bits = em(x, idx, training=False)
    assert torch.equal(y, torch.round(x)) and bits.shape == (1,)
    y2, _ = em(x, idx, loc=torch.full((1, 4), 0.25), training=False)
    assert torch.allclose(y2, torch.round(x - 0.25) + 0.25)
    multi = tfc.ContinuousIndexedEntropyModel(
        tfc.NoisyNormal, (3, 5), dict(loc=lambda i: i[..., 0], scale=lambda i: 1 + i[..., 1]), 1)
    flat = multi._flatten_indexes(torch.tensor([[[2, 4], [1, 0]]]))
    assert flat.tolist() == [[14, 5]]
