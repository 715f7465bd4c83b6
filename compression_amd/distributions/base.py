"""Minimal scalar-distribution protocol (the subset of tfp.distributions.Distribution
the entropy models rely on): batch_shape, dtype, prob/log_prob, cdf/sf and their logs,
quantile, mean.  Methods a family cannot provide raise NotImplementedError, which the
helpers in helpers.py use for their fallback chain exactly like the reference does."""
from __future__ import annotations

import math

import torch

__all__ = ["Distribution", "Normal", "Logistic", "Laplace"]


class Distribution(torch.nn.Module):
    """Batch of independent scalar distributions; event_shape is always ()."""

    def __init__(self, dtype=torch.float32):
        super().__init__()
        self._dtype = dtype

    @property
    def dtype(self):
        return self._dtype

    @property
    def event_shape(self):
        return torch.Size(())

    @property
    def batch_shape(self) -> torch.Size:
        raise NotImplementedError

    def batch_shape_tensor(self):
        return torch.tensor(list(self.batch_shape), dtype=torch.int64)

    # families override the underscored methods they support
    def _missing(self, what):
        raise NotImplementedError(f"{type(self).__name__} does not implement {what}()")

    def prob(self, x): return self._prob(torch.as_tensor(x, dtype=self.dtype))
    def log_prob(self, x): return self._log_prob(torch.as_tensor(x, dtype=self.dtype))
    def cdf(self, x): return self._cdf(torch.as_tensor(x, dtype=self.dtype))
    def log_cdf(self, x): return self._log_cdf(torch.as_tensor(x, dtype=self.dtype))
    def survival_function(self, x): return self._survival_function(torch.as_tensor(x, dtype=self.dtype))
    def log_survival_function(self, x): return self._log_survival_function(torch.as_tensor(x, dtype=self.dtype))
    def quantile(self, q): return self._quantile(torch.as_tensor(q, dtype=self.dtype))
    def mean(self): return self._mean()
    def mode(self): return self._mode()

    def _prob(self, x): return torch.exp(self._log_prob(x))
    def _log_prob(self, x): self._missing("log_prob")
    def _cdf(self, x): self._missing("cdf")
    def _log_cdf(self, x): return torch.log(self._cdf(x))
    def _survival_function(self, x): self._missing("survival_function")
    def _log_survival_function(self, x): return torch.log(self._survival_function(x))
    def _quantile(self, q): self._missing("quantile")
    def _mean(self): self._missing("mean")
    def _mode(self): self._missing("mode")


class _LocScale(Distribution):
    def __init__(self, loc, scale, dtype=torch.float32):
        super().__init__(dtype)
        self.loc = torch.as_tensor(loc, dtype=dtype)
        self.scale = torch.as_tensor(scale, dtype=dtype)
        if self.scale.device != self.loc.device:
            self.loc = self.loc.to(self.scale.device)

    @property
    def batch_shape(self):
        return torch.broadcast_shapes(self.loc.shape, self.scale.shape)

    def _z(self, x):
        return (x - self.loc.to(x.device)) / self.scale.to(x.device)

    def _mean(self):
        return self.loc.expand(self.batch_shape)

    def _mode(self):
        return self.loc.expand(self.batch_shape)


class Normal(_LocScale):
    def _log_prob(self, x):
        z = self._z(x)
        return -0.5 * z * z - torch.log(self.scale.to(x.device)) - 0.5 * math.log(2 * math.pi)

    def _cdf(self, x): return torch.special.ndtr(self._z(x))
    def _survival_function(self, x): return torch.special.ndtr(-self._z(x))
    def _log_cdf(self, x): return torch.special.log_ndtr(self._z(x))
    def _log_survival_function(self, x): return torch.special.log_ndtr(-self._z(x))

    def _quantile(self, q):
        return self.loc.to(q.device) + self.scale.to(q.device) * torch.special.ndtri(q)


class Logistic(_LocScale):
    def _log_prob(self, x):
        z = self._z(x)
        return -z - 2 * torch.nn.functional.softplus(-z) - torch.log(self.scale.to(x.device))

    def _cdf(self, x): return torch.sigmoid(self._z(x))
    def _survival_function(self, x): return torch.sigmoid(-self._z(x))
    def _log_cdf(self, x): return torch.nn.functional.logsigmoid(self._z(x))
    def _log_survival_function(self, x): return torch.nn.functional.logsigmoid(-self._z(x))

    def _quantile(self, q):
        return self.loc.to(q.device) + self.scale.to(q.device) * (torch.log(q) - torch.log1p(-q))


class Laplace(_LocScale):
    def _log_prob(self, x):
        return -torch.abs(self._z(x)) - math.log(2.0) - torch.log(self.scale.to(x.device))

    def _cdf(self, x):
        z = self._z(x)
        return 0.5 - 0.5 * torch.sign(z) * torch.expm1(-torch.abs(z))

    # log cdf / log survival function without cancellation, as the reference's Laplace (tfp) has them:
    # z < 0: log(exp(z) / 2) = z - log 2 exactly; z >= 0: log1p(-exp(-z) / 2).  The survival function is the
    # exponential of its log (tfp derives it that way), so the right tail keeps its relative accuracy; the
    # Laplace-mixture tail of continuous_base.py:298-334 is evaluated out there.
    @staticmethod
    def _log_cdf_z(z):
        return torch.where(z < 0, z - math.log(2.0), torch.log1p(-0.5 * torch.exp(-torch.abs(z))))

    def _log_cdf(self, x): return self._log_cdf_z(self._z(x))
    def _log_survival_function(self, x): return self._log_cdf_z(-self._z(x))
    def _survival_function(self, x): return torch.exp(self._log_survival_function(x))

    def _quantile(self, q):
        return self.loc.to(q.device) - self.scale.to(q.device) * torch.sign(q - 0.5) * torch.log1p(
            -2 * torch.abs(q - 0.5))
