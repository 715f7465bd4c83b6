"""Deep factorized density model (python/distributions/deep_factorized.py:50-267):
per-channel monotone MLP for the logits of the cumulative, K = len(num_filters)+1
layers with softplus-reparameterised matrices and tanh gates."""
from __future__ import annotations

import logging
import math

import torch

from . import helpers
from .base import Distribution
from .uniform_noise import UniformNoiseAdapter

__all__ = ["DeepFactorized", "NoisyDeepFactorized"]

_DEVICE_TAIL_ITERATION_CAP = 100000       # kTailMaxIters of csrc/deep_factorized_tails.hip


def _log_expm1(x: float) -> float:
    return math.log(math.expm1(x))


class DeepFactorized(Distribution):
    def __init__(self, batch_shape=(), num_filters=(3, 3), init_scale=10, dtype=torch.float32):
        super().__init__(dtype)
        self._batch_shape = torch.Size(int(s) for s in batch_shape)
        self.num_filters = tuple(int(f) for f in num_filters)
        self.init_scale = float(init_scale)
        channels = self._batch_shape.numel()
        filters = (1,) + self.num_filters + (1,)
        scale = self.init_scale ** (1 / (len(self.num_filters) + 1))
        self.matrices = torch.nn.ParameterList()
        self.biases = torch.nn.ParameterList()
        self.factors = torch.nn.ParameterList()
        for i in range(len(self.num_filters) + 1):
            init = _log_expm1(1 / scale / filters[i + 1])
            self.matrices.append(torch.nn.Parameter(
                torch.full((channels, filters[i + 1], filters[i]), init, dtype=dtype)))
            self.biases.append(torch.nn.Parameter(
                torch.rand((channels, filters[i + 1], 1), dtype=dtype) - 0.5))
            if i < len(self.num_filters):
                self.factors.append(torch.nn.Parameter(
                    torch.zeros((channels, filters[i + 1], 1), dtype=dtype)))

    @property
    def batch_shape(self):
        return self._batch_shape

    def _broadcast(self, x):
        x = x.to(self.matrices[0].device)
        return x.expand(torch.broadcast_shapes(x.shape, self._batch_shape))

    def _logits_cumulative(self, inputs):
        """deep_factorized.py:166-194: (channels, 1, batch) layout, matmul chain."""
        shape = inputs.shape
        channels = self._batch_shape.numel()
        logits = inputs.reshape(-1, 1, channels).permute(2, 1, 0)
        for i in range(len(self.num_filters) + 1):
            logits = torch.matmul(torch.nn.functional.softplus(self.matrices[i]), logits)
            logits = logits + self.biases[i]
            if i < len(self.num_filters):
                logits = logits + torch.tanh(self.factors[i]) * torch.tanh(logits)
        return logits.permute(2, 1, 0).reshape(shape)

    def _cdf(self, x): return torch.sigmoid(self._logits_cumulative(self._broadcast(x)))
    def _survival_function(self, x): return torch.sigmoid(-self._logits_cumulative(self._broadcast(x)))
    def _log_cdf(self, x): return torch.nn.functional.logsigmoid(self._logits_cumulative(self._broadcast(x)))
    def _log_survival_function(self, x):
        return torch.nn.functional.logsigmoid(-self._logits_cumulative(self._broadcast(x)))

    def _dlogits(self, x):
        x = self._broadcast(x).detach().requires_grad_(True) if not x.requires_grad else self._broadcast(x)
        with torch.enable_grad():
            logits = self._logits_cumulative(x)
            d, = torch.autograd.grad(logits.sum(), x, create_graph=torch.is_grad_enabled())
        return logits, d

    def _log_prob(self, x):
        logits, d = self._dlogits(x)
        return (torch.nn.functional.logsigmoid(logits) + torch.nn.functional.logsigmoid(-logits)
                + torch.log(d))

    def _prob(self, x):
        logits, d = self._dlogits(x)
        s = torch.sigmoid(logits)
        return s * (1 - s) * d

    def _solve_device(self, targets):
        """helpers.estimate_tails for this prior with the whole iteration on the device (csrc/deep_factorized_tails.hip:
        one workgroup per target, a thread per channel; the reference's update and stopping rule) -> [len(targets),
        *batch_shape], or None where the kernel does not apply (parameters not on a HIP device, unequal hidden widths,
        more than 1024 channels, another dtype)."""
        m0 = self.matrices[0]
        channels = self._batch_shape.numel()
        if (not m0.is_cuda or self.dtype != torch.float32 or len(set(self.num_filters)) != 1
                or self.num_filters[0] > 8 or channels > 1024):
            return None
        from .. import _lib
        from ..ops.bottleneck_ops import pack_factorized_params
        with torch.no_grad():
            params = pack_factorized_params(self)
            t = torch.tensor([float(v) for v in targets], dtype=torch.float32, device=m0.device)
            out = torch.empty((len(targets), channels), dtype=torch.float32, device=m0.device)
            iters = torch.zeros(len(targets), dtype=torch.int32, device=m0.device)
            with torch.cuda.device(m0.device):
                _lib.check(_lib.lib().tfc_deep_factorized_tails(
                    params.data_ptr(), channels, params.shape[1], len(self.num_filters) + 1, self.num_filters[0],
                    t.data_ptr(), len(targets), out.data_ptr(), iters.data_ptr(), _lib.stream_ptr()))
            if int(iters.max()) >= _DEVICE_TAIL_ITERATION_CAP:
                # the kernel's bound (the reference's loop has none): the tensor-op iteration decides instead
                logging.warning("tfc_deep_factorized_tails stopped at its iteration cap (%d); falling back to the "
                                "tensor-op iteration of helpers.estimate_tails", _DEVICE_TAIL_ITERATION_CAP)
                return None
        return out.reshape((len(targets),) + tuple(self._batch_shape))

    def _quantization_offset(self):
        solved = self._solve_device([0.0])
        if solved is not None:
            return solved[0]
        with torch.no_grad():
            dev = self.matrices[0].device
        return helpers.estimate_tails(self._logits_cumulative, 0.0, self._batch_shape, self.dtype, dev)

    def _tail(self, logits):
        solved = self._solve_device([logits])
        if solved is not None:
            return solved[0]
        dev = self.matrices[0].device
        return helpers.estimate_tails(self._logits_cumulative, logits, self._batch_shape, self.dtype, dev)

    def _lower_tail(self, tail_mass):
        return self._tail(math.log(tail_mass / 2 / (1.0 - tail_mass / 2)))

    def _upper_tail(self, tail_mass):
        return self._tail(-math.log(tail_mass / 2 / (1.0 - tail_mass / 2)))


class NoisyDeepFactorized(UniformNoiseAdapter):
    """`DeepFactorized` convolved with uniform noise (deep_factorized.py:262-267)."""

    def __init__(self, **kwargs):
        super().__init__(DeepFactorized(**kwargs))
