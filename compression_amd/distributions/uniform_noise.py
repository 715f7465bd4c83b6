"""Distributions convolved with unit-width uniform noise
(python/distributions/uniform_noise.py:54-317)."""
from __future__ import annotations

import torch

from . import helpers
from .base import Distribution, Laplace, Logistic, Normal

__all__ = ["UniformNoiseAdapter", "NoisyNormal", "NoisyLogistic", "NoisyLaplace", "MixtureSameFamily",
           "NoisyMixtureSameFamily", "NoisyNormalMixture", "NoisyLogisticMixture"]


def _logsum_expbig_minus_expsmall(big, small):
    """log(exp(big) - exp(small)), stable for big >= small (uniform_noise.py:40-51)."""
    return torch.log1p(-torch.exp(small - big)) + big


class UniformNoiseAdapter(Distribution):
    """(p * u)(x) = c(x + .5) - c(x - .5) for a base distribution with CDF c."""

    def __init__(self, base: Distribution):
        super().__init__(base.dtype)
        self.base = base

    @property
    def batch_shape(self):
        return self.base.batch_shape

    def _log_prob(self, y):
        try:
            logsf_p = self.base.log_survival_function(y + 0.5)
            logsf_m = self.base.log_survival_function(y - 0.5)
            use_sf = True
        except NotImplementedError:
            use_sf = False
        logcdf_p = self.base.log_cdf(y + 0.5)
        logcdf_m = self.base.log_cdf(y - 0.5)
        if not use_sf:
            return _logsum_expbig_minus_expsmall(logcdf_p, logcdf_m)
        # use the survival function on the right of the median (uniform_noise.py:134-156)
        cond = logsf_p < logcdf_p
        big = torch.where(cond, logsf_m, logcdf_p)
        small = torch.where(cond, logsf_p, logcdf_m)
        return _logsum_expbig_minus_expsmall(big, small)

    def _prob(self, y):
        cdf_p = self.base.cdf(y + 0.5)
        cdf_m = self.base.cdf(y - 0.5)
        try:
            sf_p = self.base.survival_function(y + 0.5)
            sf_m = self.base.survival_function(y - 0.5)
        except NotImplementedError:
            return cdf_p - cdf_m
        return torch.where(sf_p < cdf_p, sf_m - sf_p, cdf_p - cdf_m)

    def _mean(self):
        return self.base.mean()

    def _quantization_offset(self):
        return helpers.quantization_offset(self.base)

    def _lower_tail(self, tail_mass):
        return helpers.lower_tail(self.base, tail_mass)

    def _upper_tail(self, tail_mass):
        return helpers.upper_tail(self.base, tail_mass)


class NoisyNormal(UniformNoiseAdapter):
    def __init__(self, loc, scale, dtype=torch.float32):
        super().__init__(Normal(loc=loc, scale=scale, dtype=dtype))


class NoisyLogistic(UniformNoiseAdapter):
    def __init__(self, loc, scale, dtype=torch.float32):
        super().__init__(Logistic(loc=loc, scale=scale, dtype=dtype))


class NoisyLaplace(UniformNoiseAdapter):
    def __init__(self, loc, scale, dtype=torch.float32):
        super().__init__(Laplace(loc=loc, scale=scale, dtype=dtype))


class MixtureSameFamily(Distribution):
    """sum_k w_k p_k: `weight` [..., K] (probabilities, as tfp's Categorical(probs=...)), `components` a
    distribution whose batch shape ends in K — the subset of tfp.distributions.MixtureSameFamily the noisy
    mixtures below need.  No quantile (the tail helpers then solve for it, helpers.py:150-219)."""

    def __init__(self, weight, components: Distribution):
        super().__init__(components.dtype)
        self.weight = torch.as_tensor(weight, dtype=components.dtype)
        self.components = components

    @property
    def batch_shape(self):
        return torch.broadcast_shapes(self.weight.shape, self.components.batch_shape)[:-1]

    def _mix(self, values, x):
        return torch.sum(self.weight.to(x.device) * values, dim=-1)

    def _log_mix(self, log_values, x):
        return torch.logsumexp(torch.log(self.weight.to(x.device)) + log_values, dim=-1)

    def _prob(self, x): return self._mix(self.components.prob(x[..., None]), x)
    def _log_prob(self, x): return self._log_mix(self.components.log_prob(x[..., None]), x)
    def _cdf(self, x): return self._mix(self.components.cdf(x[..., None]), x)
    def _log_cdf(self, x): return self._log_mix(self.components.log_cdf(x[..., None]), x)
    def _survival_function(self, x): return self._mix(self.components.survival_function(x[..., None]), x)

    def _log_survival_function(self, x):
        return self._log_mix(self.components.log_survival_function(x[..., None]), x)

    def _mean(self):
        m = self.components.mean()
        return torch.sum(self.weight.to(m.device) * m, dim=-1)


class NoisyMixtureSameFamily(Distribution):
    """Mixture of distributions with additive i.i.d. uniform noise (uniform_noise.py:203-254): a mixture of
    the uniform-noise-adapted components; `base` is the mixture without the noise (tails are located on it)."""

    def __init__(self, weight, components: Distribution):
        super().__init__(components.dtype)
        self.components_distribution = UniformNoiseAdapter(components)
        self.mixture = MixtureSameFamily(weight, self.components_distribution)
        self.base = MixtureSameFamily(weight, components)

    @property
    def batch_shape(self):
        return self.base.batch_shape

    def _prob(self, x): return self.mixture._prob(x)
    def _log_prob(self, x): return self.mixture._log_prob(x)
    def _mean(self): return self.base.mean()

    def _quantization_offset(self):
        # the "peakiest" of the component quantization offsets (uniform_noise.py:239-245): the mixture's
        # log-probability AT each component's offset decides
        offsets = helpers.quantization_offset(self.components_distribution)
        offsets = torch.broadcast_to(offsets, tuple(self.batch_shape) + offsets.shape[-1:])
        component = torch.argmax(self.log_prob(offsets.movedim(-1, 0)), dim=0)
        return torch.gather(offsets, -1, component[..., None]).squeeze(-1)

    def _lower_tail(self, tail_mass):
        return helpers.lower_tail(self.base, tail_mass)

    def _upper_tail(self, tail_mass):
        return helpers.upper_tail(self.base, tail_mass)


class NoisyNormalMixture(NoisyMixtureSameFamily):
    """uniform_noise.py:281-298: loc / scale of the Normal components and the mixture probabilities, last axis K."""

    def __init__(self, loc, scale, weight, dtype=torch.float32):
        super().__init__(weight, Normal(loc=loc, scale=scale, dtype=dtype))


class NoisyLogisticMixture(NoisyMixtureSameFamily):
    """uniform_noise.py:301-319."""

    def __init__(self, loc, scale, weight, dtype=torch.float32):
        super().__init__(weight, Logistic(loc=loc, scale=scale, dtype=dtype))

