"""Distributions of (soft-)rounded variables (python/distributions/round_adapters.py:36-290, Appendix E of
"Universally Quantized Neural Compression"): a continuous base pushed through an ascending monotonic function
f, and the same with unit uniform noise added — the priors of the universal entropy models."""
from __future__ import annotations

import torch

from ..ops import round_ops
from . import helpers
from .base import Distribution, Normal
from .deep_factorized import DeepFactorized
from .uniform_noise import UniformNoiseAdapter

__all__ = ["MonotonicAdapter", "RoundAdapter", "NoisyRoundAdapter", "NoisyRoundedDeepFactorized",
           "NoisyRoundedNormal", "SoftRoundAdapter", "NoisySoftRoundAdapter", "NoisySoftRoundedNormal",
           "NoisySoftRoundedDeepFactorized"]


class MonotonicAdapter(Distribution):
    """f(X) for X ~ base.  With g(y) = inf {x : f(x) >= y} (the inverse of f where it has one):
    P(f(X) <= y) = P(X <= g(y)), so every cumulative is the base's at g(y) (round_adapters.py:36-163).
    No density: f(X) may be discrete."""

    invertible = True

    def __init__(self, base: Distribution):
        super().__init__(base.dtype)
        self.base = base

    @property
    def batch_shape(self):
        return self.base.batch_shape

    def transform(self, x):
        raise NotImplementedError()

    def inverse_transform(self, y):
        raise NotImplementedError()

    def _prob(self, x): self._missing("prob")
    def _log_prob(self, x): self._missing("log_prob")
    def _cdf(self, y): return self.base._cdf(self.inverse_transform(y))
    def _log_cdf(self, y): return self.base._log_cdf(self.inverse_transform(y))
    def _survival_function(self, y): return self.base._survival_function(self.inverse_transform(y))
    def _log_survival_function(self, y): return self.base._log_survival_function(self.inverse_transform(y))

    # P(X <= z) = q  <=>  P(f(X) <= f(z)) = q for an invertible f: quantiles, modes, offsets and tails map
    # through f
    def _through(self, value):
        if not self.invertible:
            raise NotImplementedError()
        return self.transform(value())

    def _quantile(self, q): return self._through(lambda: self.base._quantile(q))
    def _mode(self): return self._through(lambda: self.base._mode())
    def _quantization_offset(self): return self._through(lambda: helpers.quantization_offset(self.base))
    def _lower_tail(self, tail_mass): return self._through(lambda: helpers.lower_tail(self.base, tail_mass))
    def _upper_tail(self, tail_mass): return self._through(lambda: helpers.upper_tail(self.base, tail_mass))


class RoundAdapter(MonotonicAdapter):
    """round(X) (round_adapters.py:166-199): g(y) = ceil(y) - 1/2."""

    invertible = False

    def transform(self, x):
        return torch.round(x)

    def inverse_transform(self, y):
        return torch.ceil(y) - 0.5

    def _quantization_offset(self):
        return torch.zeros((), dtype=self.dtype)

    def _lower_tail(self, tail_mass):
        return torch.floor(helpers.lower_tail(self.base, tail_mass))

    def _upper_tail(self, tail_mass):
        return torch.ceil(helpers.upper_tail(self.base, tail_mass))


class NoisyRoundAdapter(UniformNoiseAdapter):
    """round(X) + U (round_adapters.py:202-213)."""

    def __init__(self, base: Distribution):
        super().__init__(RoundAdapter(base))


class NoisyRoundedDeepFactorized(NoisyRoundAdapter):
    def __init__(self, **kwargs):
        super().__init__(DeepFactorized(**kwargs))


class NoisyRoundedNormal(NoisyRoundAdapter):
    def __init__(self, loc, scale, dtype=torch.float32):
        super().__init__(Normal(loc=loc, scale=scale, dtype=dtype))


class SoftRoundAdapter(MonotonicAdapter):
    """soft_round(X, alpha) (round_adapters.py:231-250)."""

    def __init__(self, base: Distribution, alpha):
        super().__init__(base)
        self._alpha = alpha

    def transform(self, x):
        return round_ops.soft_round(x, self._alpha)

    def inverse_transform(self, y):
        return round_ops.soft_round_inverse(y, self._alpha)


class NoisySoftRoundAdapter(UniformNoiseAdapter):
    """soft_round(X, alpha) + U (round_adapters.py:253-265)."""

    def __init__(self, base: Distribution, alpha):
        super().__init__(SoftRoundAdapter(base, alpha))


class NoisySoftRoundedNormal(NoisySoftRoundAdapter):
    def __init__(self, loc, scale, alpha=5.0, dtype=torch.float32):
        super().__init__(Normal(loc=loc, scale=scale, dtype=dtype), alpha)


class NoisySoftRoundedDeepFactorized(NoisySoftRoundAdapter):
    def __init__(self, alpha=5.0, **kwargs):
        super().__init__(DeepFactorized(**kwargs), alpha)
