"""CPU tier: soft rounding (python/ops/round_ops.py:46-133, layers/soft_round.py) and the (soft-)round
adapters (python/distributions/round_adapters.py) — the reference's round_ops_test.py, soft_round_test.py and
round_adapters_test.py cases."""
import numpy as np
import pytest
import torch

import compression_amd as tfc
from compression_amd.distributions import helpers


# ------------------------------------------------------------------ round_ops_test.py:25-91
def test_soft_round_limits():
    x = torch.linspace(-2.0, 2.0, 50)
    assert torch.allclose(tfc.soft_round(x, alpha=1e-13), x)
    assert torch.equal(tfc.soft_round_inverse(x, alpha=1e-13), x)
    for offset in range(-5, 5):
        x = torch.linspace(offset - 0.499, offset + 0.499, 100)
        assert torch.allclose(tfc.soft_round(x, alpha=2000.0), torch.round(x), atol=0.02)
        x = torch.linspace(offset + 0.001, offset + 0.999, 100)
        assert torch.allclose(tfc.soft_round_inverse(x, alpha=5000.0), torch.ceil(x) - 0.5, atol=0.001)
        assert torch.allclose(tfc.soft_round_conditional_mean(x, alpha=5000.0), torch.round(x), atol=0.001)


def test_soft_round_inverse_is_the_inverse():
    x = torch.tensor([-1.25, -0.75, 0.75, 1.25])
    assert torch.allclose(tfc.soft_round_inverse(tfc.soft_round(x, alpha=2.0), alpha=2.0), x)


@pytest.mark.parametrize("alpha", [0.0, 1e-6, 1e-2, 5.0, 1e6])
def test_soft_round_values_and_gradients_are_finite(alpha):
    x = torch.linspace(0.0, 1.0, 11, requires_grad=True)         # integers and half-integers included
    y = tfc.soft_round(x, alpha=alpha)
    dy, = torch.autograd.grad(y.sum(), x)
    assert torch.isfinite(y).all() and torch.isfinite(dy).all()
    x = torch.linspace(-0.5, 0.5, 11, requires_grad=True)
    y = tfc.soft_round_inverse(x, alpha=alpha)
    dy, = torch.autograd.grad(y.sum(), x)
    assert torch.isfinite(y).all()
    finite = torch.isfinite(dy)
    if alpha > 15:
        finite[5] = True            # extremely steep at 0 for large alpha: a non-finite slope is allowed there
    assert finite.all()


# ------------------------------------------------------------------ soft_round_test.py
def test_soft_round_layers():
    x = torch.linspace(-5.0, 5.0, 50)
    y = tfc.SoftRound(alpha=3.0)(x)
    assert torch.allclose(tfc.SoftRound(alpha=3.0, inverse=True)(y), x, atol=1e-5)
    assert torch.allclose(tfc.SoftRoundConditionalMean(alpha=5000.0)(x[1:-1] + 0.013), torch.round(x[1:-1] + 0.013),
                          atol=0.001)
    for alpha in (0.0, 1e-3):
        assert torch.allclose(tfc.SoftRound(alpha=alpha)(x), x, atol=1e-5)


# ------------------------------------------------------------------ round_adapters_test.py:39-140
def _bases():
    D = tfc.distributions
    return {
        "softround_deepfactorized": (lambda d: D.SoftRoundAdapter(d, alpha=5.0), lambda: D.DeepFactorized()),
        "softround_logistic": (lambda d: D.SoftRoundAdapter(d, alpha=5.0), lambda: D.Logistic(loc=10.3, scale=1.5)),
        "softround_normal": (lambda d: D.SoftRoundAdapter(d, alpha=4.0), lambda: D.Normal(loc=10.4, scale=1.5)),
        "noisysoftround_deepfactorized": (lambda d: D.NoisySoftRoundAdapter(d, alpha=5.0), lambda: D.DeepFactorized()),
        "noisysoftround_logistic": (lambda d: D.NoisySoftRoundAdapter(d, alpha=5.0),
                                    lambda: D.Logistic(loc=10.0, scale=1.5)),
        "noisysoftround_normal": (lambda d: D.NoisySoftRoundAdapter(d, alpha=5.0), lambda: D.Normal(loc=10.0, scale=1.5)),
        "round_deepfactorized": (D.RoundAdapter, lambda: D.DeepFactorized(init_scale=1.0)),
        "round_logistic": (D.RoundAdapter, lambda: D.Logistic(loc=1.5, scale=1.5)),
        "round_normal": (D.RoundAdapter, lambda: D.Normal(loc=1.5, scale=1.5)),
        "noisyround_deepfactorized": (D.NoisyRoundAdapter, lambda: D.DeepFactorized(init_scale=1.0)),
        "noisyround_logistic": (D.NoisyRoundAdapter, lambda: D.Logistic(loc=1.5, scale=1.5)),
        "noisyround_normal": (D.NoisyRoundAdapter, lambda: D.Normal(loc=1.5, scale=1.5)),
    }


@pytest.mark.parametrize("name", sorted(_bases()))
def test_adapter_tails(name):
    adapter, base = _bases()[name]
    torch.manual_seed(0)
    dist = adapter(base())
    lower, upper = dist._lower_tail(2 ** -8), dist._upper_tail(2 ** -8)
    try:
        left = dist.cdf(lower)
    except NotImplementedError:
        left = dist.base.cdf(lower)            # the base as a proxy for the tail mass
    try:
        right = dist.survival_function(upper)
    except NotImplementedError:
        right = dist.base.survival_function(upper)
    assert float(left.detach()) <= 2 ** -8 and float(right.detach()) <= 2 ** -8 and float(upper) > float(lower)


@pytest.mark.parametrize("base", ["Logistic", "Normal"])
def test_soft_round_adapter_mode_and_quantile(base):
    D = tfc.distributions
    dist = D.SoftRoundAdapter(getattr(D, base)(loc=10.0, scale=1.5), alpha=5.0)
    assert abs(float(dist.cdf(dist.mode())) - 0.5) < 1e-5
    assert abs(float(dist.cdf(dist.quantile(0.75))) - 0.75) < 1e-5


def test_adapters_lacking_an_inverse():
    D = tfc.distributions
    dist = D.RoundAdapter(D.Logistic(loc=1.5, scale=1.5))
    for what in (dist.mode, lambda: dist.quantile(0.75)):
        with pytest.raises(NotImplementedError):
            what()

    class NonInvertible(D.MonotonicAdapter):
        invertible = False

        def transform(self, x):
            return torch.ceil(x)

        def inverse_transform(self, y):
            return torch.floor(y)

    dist = NonInvertible(D.Normal(loc=1.5, scale=1.5))
    for what in (lambda: dist._lower_tail(0.01), lambda: dist._upper_tail(0.01)):
        with pytest.raises(NotImplementedError):
            what()


# ------------------------------------------------------------------ round_adapters_test.py:143-250
def _log_prob_gradient_is_bounded(dist, values):
    x = torch.tensor(values, requires_grad=True)
    p = dist.log_prob(x)
    idx = (p < -32.0).detach()
    dx, = torch.autograd.grad(torch.clamp(p, min=-32.0).sum(), x)
    assert torch.equal(dx[idx], torch.zeros_like(dx[idx]))
    assert torch.isfinite(dx).all(), f"dx has a non-finite value: {dx}"


def test_noisy_soft_rounded_deep_factorized():
    torch.manual_seed(0)
    df = tfc.NoisySoftRoundedDeepFactorized(init_scale=1e-3)
    assert torch.allclose(df.prob(torch.linspace(-1.0, 1.0, 10)),
                          torch.tensor([0, 0, 0, 1, 1, 1, 1, 0, 0, 0.0]), atol=1e-4)
    _log_prob_gradient_is_bounded(tfc.NoisySoftRoundedDeepFactorized(), [0.0, 1.0, 2.0, 1e3])


@pytest.mark.parametrize("dist_cls", ["NoisyRoundedNormal", "NoisySoftRoundedNormal"])
def test_noisy_rounded_location_scale(dist_cls):
    cls = getattr(tfc, dist_cls)
    dist = cls(loc=3.0, scale=5.0)
    assert tuple(dist.batch_shape) == () and tuple(dist.event_shape) == ()
    assert tuple(cls(loc=[3.0, 2.0], scale=5.0).batch_shape) == (2,)
    loc = torch.tensor(1.0, requires_grad=True)
    log_scale = torch.tensor(0.0, requires_grad=True)
    torch.manual_seed(0)
    loss = -cls(loc=loc, scale=torch.exp(log_scale)).log_prob(torch.randn(20)).mean()
    grads = torch.autograd.grad(loss, [loc, log_scale])
    assert all(torch.isfinite(g) for g in grads)
    x = torch.linspace(4.0, 6.0, 10)
    assert torch.allclose(cls(loc=5.0, scale=1e-7).prob(x), torch.tensor([0, 0, 0, 1, 1, 1, 1, 0, 0, 0.0]), atol=1e-5)
    dist = cls(loc=10.0, scale=1.5)
    assert float(dist._upper_tail(2 ** -8)) > float(dist._lower_tail(2 ** -8))
    dist = cls(loc=1.0, scale=2.0)
    for what in (dist.mode, lambda: dist.quantile(0.5), lambda: dist.survival_function(0.5)):
        with pytest.raises(NotImplementedError):
            what()
    if dist_cls == "NoisySoftRoundedNormal":
        _log_prob_gradient_is_bounded(cls(loc=0.0, scale=1.0), [0.0, 1.0, 2.0, 1e3])


def test_universal_model_with_a_soft_rounded_prior():
    """The priors above are what the universal entropy models are trained with: rate estimates of the
    training and the evaluation call agree for a smooth source."""
    torch.manual_seed(0)
    prior = tfc.NoisySoftRoundedNormal(loc=0.0, scale=torch.full((8,), 3.0), alpha=2.0)
    em = tfc.UniversalBatchedEntropyModel(prior, coding_rank=2, compression=False, num_noise_levels=15)
    y = tfc.soft_round(torch.randn(4, 500, 8) * 3.0, 2.0)
    _, bits_t = em(y, training=True)
    _, bits_e = em(y, training=False)
    assert torch.isfinite(bits_t).all() and abs(float(bits_t.mean() / bits_e.mean()) - 1) < 0.03
