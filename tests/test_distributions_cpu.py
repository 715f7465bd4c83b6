"""CPU tier: the noisy mixture priors (python/distributions/uniform_noise.py:203-319) — the reference's own
MixtureTest cases (uniform_noise_test.py:105-182) — and the universal models' tests that use them."""
import math

import numpy as np
import pytest
import torch

import compression_amd as tfc
from compression_amd.distributions import helpers

MIXTURES = [tfc.NoisyNormalMixture, tfc.NoisyLogisticMixture]


@pytest.mark.parametrize("dist_cls", MIXTURES)
def test_mixture_shapes(dist_cls):
    dist = dist_cls(loc=[3.0, -3.0], scale=[5.0, 2.5], weight=[0.3, 0.7])
    assert tuple(dist.batch_shape) == () and tuple(dist.event_shape) == ()
    dist = dist_cls(loc=[[3.0, -3.0], [2.0, -2.0]], scale=[5.0, 2.5], weight=[0.3, 0.7])
    assert tuple(dist.batch_shape) == (2,) and tuple(dist.event_shape) == ()
    assert dist.log_prob(torch.zeros(7, 2)).shape == (7, 2)


@pytest.mark.parametrize("dist_cls", MIXTURES)
def test_mixture_parameters_receive_gradients(dist_cls):
    loc = torch.ones(2, requires_grad=True)
    log_scale = torch.zeros(2, requires_grad=True)
    logit_weight = torch.tensor([0.3, 0.7], requires_grad=True)
    dist = dist_cls(loc=loc, scale=torch.exp(log_scale), weight=torch.softmax(logit_weight, -1))
    torch.manual_seed(0)
    loss = -dist.log_prob(torch.randn(20)).mean()
    grads = torch.autograd.grad(loss, [loc, log_scale, logit_weight])
    assert all(g is not None and torch.isfinite(g).all() and g.abs().sum() > 0 for g in grads)


@pytest.mark.parametrize("dist_cls", MIXTURES)
def test_mixture_of_uniforms_is_the_limit(dist_cls):
    dist = dist_cls(loc=[2.5, -1.0], scale=[1e-7, 1e-7], weight=[0.3, 0.7])
    mean = dist.components_distribution.mean()
    for k, w in ((0, 0.3), (1, 0.7)):
        x = torch.linspace(float(mean[k]) - 1, float(mean[k]) + 1, 10)
        assert torch.allclose(dist.prob(x), torch.tensor([0, 0, 0, w, w, w, w, 0, 0, 0]), atol=1e-6)


@pytest.mark.parametrize("dist_cls", MIXTURES)
def test_mixture_tails_and_offset(dist_cls):
    dist = dist_cls(loc=[5.4, 8.6], scale=[1.4, 2.0], weight=[0.6, 0.4])
    lower, upper = helpers.lower_tail(dist, 2 ** -8), helpers.upper_tail(dist, 2 ** -8)
    assert upper > lower
    # the tails cut 2^-9 of the noise-free mixture off either side
    assert abs(float(dist.base.cdf(lower)) - 2 ** -9) < 2e-4
    assert abs(float(dist.base.survival_function(upper)) - 2 ** -9) < 2e-4
    assert abs(float(helpers.quantization_offset(dist)) - 0.4) < 1e-6      # the peakiest mode is 5.4
    for what in (dist.mode, lambda: dist.quantile(0.5), lambda: dist.survival_function(0.5)):
        with pytest.raises(NotImplementedError):
            what()


@pytest.mark.parametrize("dist_cls", MIXTURES)
def test_mixture_is_stable_at_zero_scale(dist_cls):
    dist = dist_cls(loc=[0.0, 0.0], scale=[0.0, 0.0], weight=[0.5, 0.5])
    assert torch.allclose(dist.prob([0.0]), torch.tensor([1.0]))
    assert torch.allclose(dist.prob([1.0]), torch.tensor([0.0]))


def test_mixture_is_a_pmf_on_the_integer_grid():
    dist = tfc.NoisyNormalMixture(loc=[-2.3, 4.1], scale=[1.5, 0.7], weight=[0.25, 0.75])
    x = torch.arange(-40.0, 41.0)
    assert abs(float(dist.prob(x).sum()) - 1) < 1e-5
    assert torch.allclose(dist.log_prob(x).exp(), dist.prob(x), atol=1e-7)


def _mixture_model(expected_grads):
    """universal_test.py:377-436: three index dimensions drive the two locations and the weights."""
    return tfc.UniversalIndexedEntropyModel(
        tfc.NoisyLogisticMixture, index_ranges=(10, 10, 5),
        parameter_fns=dict(loc=lambda i: i[..., 0:2] - 5, scale=lambda _: 1.0,
                           weight=lambda i: torch.softmax((i[..., 2:3] - 2) * torch.tensor([-1.0, 1.0]), -1)),
        coding_rank=2, expected_grads=expected_grads)


def test_universal_indexed_model_with_a_mixture_prior_expected_grads_or_not():
    """universal_test.py:377-414 (fewer symbols): same bits either way, perturbation within 1/2."""
    torch.manual_seed(0)
    x = torch.randn(3, 2000, 16)
    indexes = torch.floor(10 * torch.rand(3, 2000, 16, 3))
    torch.manual_seed(1)
    x_hat, bits_expected = _mixture_model(True)(x, indexes)
    assert (x - x_hat).abs().max() <= 0.5
    torch.manual_seed(1)
    x_hat, bits_plain = _mixture_model(False)(x, indexes)
    assert (x - x_hat).abs().max() <= 0.5
    assert torch.allclose(bits_plain, bits_expected, rtol=1e-3)


def test_universal_indexed_model_with_a_mixture_prior_gives_gradients():
    """universal_test.py:416-436: gradients of bits reach the bottleneck and the indexes; the perturbed
    bottleneck does not depend on the indexes."""
    torch.manual_seed(0)
    x = torch.randn(3, 500, 16, requires_grad=True)
    indexes = (10 * torch.rand(3, 500, 16, 3)).requires_grad_(True)
    x2, bits = _mixture_model(True)(x, indexes)
    gx, gi = torch.autograd.grad(bits.sum(), [x, indexes], retain_graph=True)
    assert torch.isfinite(gx).all() and gx.abs().sum() > 0 and torch.isfinite(gi).all() and gi.abs().sum() > 0
    assert torch.autograd.grad(x2.sum(), [x], retain_graph=True)[0] is not None
    assert torch.autograd.grad(x2.sum(), [indexes], allow_unused=True)[0] is None


def test_laplace_tail_mass_for_large_and_small_inputs():
    """universal_test.py:94-119 (without the range-coding tables, which need the device): with the tail
    the cost of a far outlier is |x| nats — in float32 this needs the Laplace log cdf / log survival function
    evaluated without cancellation — and inputs near the prior are unaffected."""
    torch.manual_seed(0)
    prior = tfc.NoisyDeepFactorized(batch_shape=(1,))
    em1 = tfc.UniversalBatchedEntropyModel(prior, coding_rank=1, laplace_tail_mass=1e-3)
    em2 = tfc.UniversalBatchedEntropyModel(prior, coding_rank=1)
    x = torch.tensor([1e3, 1e4, 1e5, 1e6])
    _, bits = em1(x[..., None])
    assert torch.allclose(bits.detach(), x.abs() / math.log(2.0), rtol=0.01)
    x = torch.linspace(-10.0, 10.0, 50)
    torch.manual_seed(1)
    _, bits1 = em1(x[..., None])
    torch.manual_seed(1)
    _, bits2 = em2(x[..., None])
    assert torch.allclose(bits1, bits2, rtol=0.01, atol=0.05)


# ------------------------------------------------------------------ helpers_test.py
def test_estimate_tails_terminates_on_nan_and_on_a_perfect_initial_guess():
    helpers.estimate_tails(lambda x: torch.tanh(x) * float("nan"), target=0.5, shape=(), dtype=torch.float32)
    # the initial guess is zero: no zero crossing would ever start the count
    helpers.estimate_tails(torch.tanh, target=0.0, shape=(), dtype=torch.float32)


@pytest.mark.parametrize("family,loc,scale", [("Laplace", -2.0, 5.0), ("Logistic", -3.0, 1.0), ("Normal", 3.0, 5.0)])
def test_location_scale_offsets_and_tails(family, loc, scale):
    dist = getattr(tfc.distributions, family)(loc=loc, scale=scale)
    assert float(helpers.quantization_offset(dist)) == 0.0          # decimal part of the mode
    assert float(helpers.upper_tail(dist, 2 ** -8)) > float(helpers.lower_tail(dist, 2 ** -8))
    assert abs(float(dist.cdf(helpers.lower_tail(dist, 2 ** -8))) - 2 ** -9) < 1e-5
    shifted = getattr(tfc.distributions, family)(loc=1.4, scale=3.0)
    assert abs(float(helpers.quantization_offset(shifted)) - 0.4) < 1e-6


@pytest.mark.parametrize("cls", ["DeepFactorized", "NoisyDeepFactorized"])
def test_deep_factorized_tails_are_in_order(cls):
    torch.manual_seed(0)
    dist = getattr(tfc, cls)(batch_shape=[10])
    assert (helpers.upper_tail(dist, 2 ** -8) - helpers.lower_tail(dist, 2 ** -8) > 0).all()


# ------------------------------------------------------------------ deep_factorized_test.py
def test_deep_factorized_shapes_and_defaults():
    df = tfc.DeepFactorized()
    assert tuple(df.batch_shape) == () and tuple(df.event_shape) == ()
    assert df.num_filters == (3, 3) and df.init_scale == 10
    df = tfc.DeepFactorized(batch_shape=(4, 3))
    assert tuple(df.batch_shape) == (4, 3)
    noisy = tfc.NoisyDeepFactorized(num_filters=(2, 3, 4))
    assert noisy.base.num_filters == (2, 3, 4) and tuple(noisy.batch_shape) == ()
    noisy.prob(torch.randn(10))
    tfc.NoisyDeepFactorized(batch_shape=(4, 3)).prob(torch.randn(10, 4, 3))


@pytest.mark.parametrize("method", ["prob", "log_prob", "cdf", "log_cdf", "survival_function",
                                    "log_survival_function"])
def test_deep_factorized_methods(method):
    # without hidden units the density collapses to a logistic distribution
    torch.manual_seed(0)
    df = tfc.DeepFactorized(num_filters=(), init_scale=1)
    logistic = tfc.distributions.Logistic(loc=-df.biases[0].detach().reshape(()), scale=1.0)
    x = torch.linspace(-5.0, 5.0, 20)
    assert torch.allclose(getattr(df, method)(x), getattr(logistic, method)(x), atol=1e-6, rtol=1e-5)
    # broadcasting: [4, 5, 1, 1] against batch shape (2, 3)
    df = tfc.DeepFactorized(batch_shape=(2, 3))
    assert getattr(df, method)(torch.linspace(-5.0, 5.0, 20).reshape(4, 5, 1, 1)).shape == (4, 5, 2, 3)


def test_noisy_deep_factorized_special_cases_and_gradients():
    torch.manual_seed(0)
    df = tfc.NoisyDeepFactorized()
    loss = -df.log_prob(torch.randn(20)).mean()
    grads = torch.autograd.grad(loss, list(df.parameters()))
    assert len(grads) == 8 and all(g is not None for g in grads)
    df = tfc.NoisyDeepFactorized(num_filters=(), init_scale=1)
    logistic = tfc.distributions.Logistic(loc=-df.base.biases[0].detach().reshape(()), scale=1.0)
    x = torch.linspace(-5.0, 5.0, 20)
    assert torch.allclose(df.prob(x), logistic.cdf(x + 0.5) - logistic.cdf(x - 0.5), atol=1e-6)
    df = tfc.NoisyDeepFactorized(init_scale=1e-3)
    assert torch.allclose(df.prob(torch.linspace(-1.0, 1.0, 10)), torch.tensor([0, 0, 0, 1, 1, 1, 1, 0, 0, 0.0]),
                          atol=1e-4)
    df = tfc.NoisyDeepFactorized()
    assert float(helpers.upper_tail(df, 2 ** -8)) > float(helpers.lower_tail(df, 2 ** -8))
    for what in (df.mode, df.mean, lambda: df.quantile(0.5), lambda: df.survival_function(0.5)):
        with pytest.raises(NotImplementedError):
            what()
