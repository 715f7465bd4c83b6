"""Base class of the continuous entropy models
(python/entropy_models/continuous_base.py:30-370): holds the integer range-coding
tables (`cdf`, `cdf_offset`) and builds them from a prior."""
from __future__ import annotations

import abc
import logging

import torch

from .. import _lib
from ..distributions import helpers, uniform_noise
from ..ops import gen_ops

__all__ = ["ContinuousEntropyModelBase"]


class ContinuousEntropyModelBase(torch.nn.Module, metaclass=abc.ABCMeta):
    def __init__(self, coding_rank=None, compression=False, stateless=False, expected_grads=False,
                 tail_mass=2 ** -8, bottleneck_dtype=None, laplace_tail_mass=0):
        super().__init__()
        self._prior = None
        self._coding_rank = int(coding_rank)
        self._compression = bool(compression)
        self._stateless = bool(stateless)
        self._expected_grads = bool(expected_grads)
        self._tail_mass = float(tail_mass)
        self._bottleneck_dtype = bottleneck_dtype or torch.get_default_dtype()
        self._laplace_tail_mass = laplace_tail_mass
        if self.coding_rank < 0:
            raise ValueError("`coding_rank` must be at least 0.")
        if not 0 < self.tail_mass < 1:
            raise ValueError("`tail_mass` must be between 0 and 1.")

    def _check_compression(self):
        if not self.compression:
            raise RuntimeError(
                "For range coding, the entropy model must be instantiated with `compression=True`.")

    @property
    def prior(self):
        if self._prior is None:
            raise RuntimeError(
                "This entropy model doesn't hold a reference to its prior distribution. This can "
                "happen depending on how it is instantiated, (e.g., if it is unserialized).")
        return self._prior

    @prior.deleter
    def prior(self):
        self._prior = None

    @property
    def cdf(self):
        self._check_compression()
        return self._cdf

    @property
    def cdf_offset(self):
        self._check_compression()
        return self._cdf_offset

    bottleneck_dtype = property(lambda self: self._bottleneck_dtype)
    expected_grads = property(lambda self: self._expected_grads)
    laplace_tail_mass = property(lambda self: self._laplace_tail_mass)
    coding_rank = property(lambda self: self._coding_rank)
    compression = property(lambda self: self._compression)
    stateless = property(lambda self: self._stateless)
    tail_mass = property(lambda self: self._tail_mass)

    @property
    def range_coder_precision(self):
        return int(-self.cdf[0])

    def _init_compression(self, cdf, cdf_offset, cdf_shapes):
        """continuous_base.py:167-215.  Tables are persistent buffers ("cdf",
        "cdf_offset" in the state dict) unless stateless; they must be stored, never
        regenerated on the receiving side."""
        if not ((cdf is None) == (cdf_offset is None) == (cdf_shapes is not None)):
            raise ValueError("Either both `cdf` and `cdf_offset`, or `cdf_shapes` must be provided.")
        if cdf_shapes is not None:
            if self.stateless:
                raise ValueError("With `stateless=True`, can't provide `cdf_shapes`.")
            cdf_shapes = tuple(map(int, cdf_shapes))
            if len(cdf_shapes) != 2:
                raise ValueError("`cdf_shapes` must have two elements.")
            cdf = torch.zeros(cdf_shapes[:1], dtype=torch.int32)
            cdf_offset = torch.zeros(cdf_shapes[1:], dtype=torch.int32)
        cdf = torch.as_tensor(cdf).to(torch.int32)
        cdf_offset = torch.as_tensor(cdf_offset).to(torch.int32)
        if self.stateless:
            self._cdf, self._cdf_offset = cdf, cdf_offset
        else:
            self.register_buffer("_cdf", cdf)
            self.register_buffer("_cdf_offset", cdf_offset)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # table sizes are data dependent (validate_shape=False in the reference)
        for name in ("_cdf", "_cdf_offset"):
            key = prefix + name
            if key in state_dict and hasattr(self, name) and name in self._buffers:
                self._buffers[name] = torch.empty_like(state_dict[key])
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _build_tables(self, prior, precision, offset=None):
        """continuous_base.py:217-296: tails -> integer support [minima, maxima] per
        scalar distribution -> PMF samples (+ overflow mass) -> pmf_to_quantized_cdf per
        row -> ragged 1-D table with NEGATIVE precision headers (escape coding on)."""
        precision = int(precision)
        dtype = prior.dtype
        with torch.no_grad():
            offset_t = torch.zeros((), dtype=dtype) if offset is None else torch.as_tensor(offset).to(dtype)
            lower = helpers.lower_tail(prior, self.tail_mass)
            upper = helpers.upper_tail(prior, self.tail_mass)
            dev = lower.device
            offset_t = offset_t.to(dev)
            minima = torch.floor(lower - offset_t).to(torch.int32)
            maxima = torch.ceil(upper - offset_t).to(torch.int32)
            pmf_start = minima.to(dtype) + offset_t
            pmf_length = maxima - minima + 1
            max_length = int(pmf_length.max())
            if max_length > 2048:
                logging.warning(
                    "Very wide PMF with %d elements may lead to out of memory issues. Consider "
                    "priors with smaller variance, or increasing `tail_mass` parameter.", max_length)
            samples = torch.arange(max_length, dtype=dtype, device=dev)
            samples = samples.reshape((-1,) + pmf_length.dim() * (1,)) + pmf_start
        pmf = prior.prob(samples).detach()
        pmf_shape = pmf.shape[1:]
        num_pmfs = int(torch.Size(pmf_shape).numel())
        cdf_offset = minima.expand(pmf_shape).reshape(num_pmfs)
        # All rows in ONE launch (include/tfc_hip.h tfc_build_tables): overflow mass, PmfToQuantizedCdf and the ragged
        # [-precision, cdf...] layout per row on the device, one read-back for the whole model (round 4: a launch, a
        # device sum and a read-back per row — 192 of each for bls2017)
        device = _lib.require_device()
        pmf = pmf.reshape(max_length, num_pmfs).t().to(device)
        lengths = pmf_length.expand(pmf_shape).reshape(num_pmfs).to(device, torch.int32).contiguous()
        inside = torch.arange(max_length, device=device)[None, :] < lengths[:, None]
        overflow = None
        if dtype != torch.float32:
            # the reference sums in the prior's dtype and casts afterwards (continuous_base.py:277-279); a float32 prior's
            # sum is the kernel's (fixed order, csrc/pmf_to_cdf.hip)
            mass = torch.where(inside, pmf, torch.zeros((), dtype=pmf.dtype, device=device)).sum(1)
            overflow = torch.clamp(1 - mass, min=0).to(torch.float32).contiguous()
        pmf = pmf.to(torch.float32).contiguous()
        ends = torch.cumsum(lengths.to(torch.int64) + 3, 0)
        bad = inside & ~(torch.isfinite(pmf) & (pmf >= 0))
        total, any_bad = (int(v) for v in torch.stack([ends[-1], bad.any().to(torch.int64)]).cpu())
        if any_bad:
            raise ValueError(
                f"`pmf` has non-finite or negative element: {pmf[bad][0].item()}. Please check for numerical "
                "problems in the probability computation.")
        offsets = (ends - (lengths.to(torch.int64) + 3)).contiguous()
        cdf = torch.empty(total, dtype=torch.int32, device=device)
        with torch.cuda.device(device):
            _lib.check(_lib.lib().tfc_build_tables_overflow(
                pmf.data_ptr(), num_pmfs, max_length, lengths.data_ptr(), offsets.data_ptr(), max_length, precision,
                overflow.data_ptr() if overflow is not None else None, cdf.data_ptr(), _lib.stream_ptr()))
        return cdf.cpu(), cdf_offset.cpu()

    def _log_prob(self, prior, bottleneck_perturbed):
        """continuous_base.py:298-334 (optional Laplace-mixture tail for stability)."""
        x = bottleneck_perturbed.to(prior.dtype)
        ltm = self.laplace_tail_mass
        if not (torch.is_tensor(ltm) or ltm > 0):
            return prior.log_prob(x)
        ltm_t = torch.as_tensor(ltm, dtype=prior.dtype, device=x.device)
        laplace = uniform_noise.NoisyLaplace(loc=0.0, scale=1.0, dtype=prior.dtype)
        probs = (1 - ltm_t) * prior.prob(x) + ltm_t * laplace.prob(x)
        small = probs < 1e-10
        mixed = torch.where(small, torch.log(ltm_t) + laplace.log_prob(x),
                            torch.log(torch.clamp(probs, min=1e-10)))
        if torch.is_tensor(ltm):
            return torch.where(ltm_t > 0, mixed, prior.log_prob(x))
        return mixed

    @abc.abstractmethod
    def get_config(self):
        if self.stateless or not self.compression:
            raise RuntimeError(
                "Serializing entropy models with `compression=False` or `stateless=True` is not "
                "supported.")
        return dict(
            coding_rank=self.coding_rank, compression=True, stateless=False,
            expected_grads=self.expected_grads, tail_mass=self.tail_mass,
            cdf_shapes=(int(self.cdf.shape[0]), int(self.cdf_offset.shape[0])),
            bottleneck_dtype=str(self.bottleneck_dtype).replace("torch.", ""),
            laplace_tail_mass=float(self.laplace_tail_mass),
        )

    def get_weights(self):
        return [b.detach().cpu().numpy() for b in self.buffers()]

    def set_weights(self, weights):
        bufs = list(self._buffers.keys())
        if len(weights) != len(bufs):
            raise ValueError(
                f"`set_weights` expects a list of {len(bufs)} arrays, received {len(weights)}.")
        for name, w in zip(bufs, weights):
            old = self._buffers[name]
            self._buffers[name] = torch.as_tensor(w).to(old.dtype)
