"""Distributions convolved with unit-width uniform noise
(python/distributions/uniform_noise.py:54-317)."""
from __future__ import annotations

import torch

from . import helpers
from .base import Distribution, Laplace, Logistic, Normal

__all__ = ["UniformNoiseAdapter", "NoisyNormal", "NoisyLogistic", "NoisyLaplace"]


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
