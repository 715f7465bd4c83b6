"""Reparameterisations of layer weights (python/layers/parameters.py:69-269).
Host-side float math on the weights, once per forward; no HIP kernel."""
from __future__ import annotations

import math

import torch

from ..ops import math_ops

__all__ = ["rdft_from_kernel", "kernel_from_rdft", "gdn_reparam_init", "gdn_reparam_value", "Parameter",
           "RDFTParameter", "GDNParameter"]


def rdft_from_kernel(kernel: torch.Tensor):
    """HWIO kernel -> (real, imag) of its normalised 2-D RDFT, shape [I, O, kh, kw//2+1]
    (RDFTParameter.__init__, parameters.py:85-127)."""
    kh, kw = kernel.shape[:2]
    spec = torch.fft.rfft2(kernel.permute(2, 3, 0, 1)) / math.sqrt(kh * kw)
    return spec.real.contiguous(), spec.imag.contiguous()


def kernel_from_rdft(real: torch.Tensor, imag: torch.Tensor, support):
    """Inverse of the above (RDFTParameter.__call__, parameters.py:140-170)."""
    kh, kw = support
    real, imag = real.float(), imag.float()       # no half-precision complex math
    spec = torch.complex(real, imag) * math.sqrt(kh * kw)
    return torch.fft.irfft2(spec, s=(kh, kw)).permute(2, 3, 0, 1)


def gdn_reparam_init(initial_value: torch.Tensor, offset: float = 2 ** -18):
    """variable = sqrt(max(v + pedestal, pedestal)) (GDNParameter.__init__, :231-240)."""
    pedestal = offset ** 2
    return torch.sqrt(torch.clamp(initial_value + pedestal, min=pedestal))


def gdn_reparam_value(variable: torch.Tensor, minimum: float = 0.0, offset: float = 2 ** -18):
    """value = lower_bound(variable, sqrt(minimum + pedestal))^2 - pedestal (:243-253);
    lower_bound carries the `identity_if_towards` gradient rule."""
    pedestal = offset ** 2
    bound = (minimum + pedestal) ** 0.5
    return torch.square(math_ops.lower_bound(variable, bound)) - pedestal


class Parameter(torch.nn.Module):
    """A layer parameter that is a function of stored variables (parameters.py:30-55): calling it gives the
    value.  Layers take such an object wherever they take a tensor or a callable."""

    def forward(self, compute_dtype=None):
        raise NotImplementedError()

    def get_config(self):
        return {}

    def get_weights(self):
        return [p.detach().cpu().numpy() for p in self.parameters()]

    def set_weights(self, weights):
        own = list(self.parameters())
        if len(weights) != len(own):
            raise ValueError(f"set_weights() expects a list of {len(own)} arrays, received {len(weights)}.")
        with torch.no_grad():
            for p, w in zip(own, weights):
                p.copy_(torch.as_tensor(w, dtype=p.dtype))


class RDFTParameter(Parameter):
    """RDFT reparameterisation of a convolution kernel (parameters.py:71-183): the variables are the real and
    imaginary parts of the kernel's normalised real-input DFT over its spatial axes; kernels of rank 3, 4, 5
    ([*support, in, out])."""

    def __init__(self, initial_value, shape=None, dtype=None):
        super().__init__()
        if initial_value is None:
            if shape is None:
                raise ValueError("If initial_value is None, shape must be specified.")
            initial_value = torch.zeros(tuple(shape), dtype=dtype or torch.float32)
        else:
            initial_value = torch.as_tensor(initial_value, dtype=dtype)
        self._shape = tuple(initial_value.shape)
        self._dtype = initial_value.dtype
        rank = len(self._shape)
        if rank not in (3, 4, 5):
            raise ValueError(f"Expected kernel tensor of rank 3, 4, or 5; received shape {self._shape}.")
        n = rank - 2
        spec = torch.fft.rfftn(initial_value.movedim((-2, -1), (0, 1)).float(), dim=tuple(range(2, 2 + n)))
        spec = spec / math.sqrt(math.prod(self._shape[:-2]))
        self.real = torch.nn.Parameter(spec.real.contiguous())
        self.imag = torch.nn.Parameter(spec.imag.contiguous())

    dtype = property(lambda self: self._dtype)
    shape = property(lambda self: self._shape)

    def forward(self, compute_dtype=None):
        real, imag = self.real, self.imag
        if compute_dtype in (torch.bfloat16, torch.float16) or compute_dtype is None:
            real, imag = real.float(), imag.float()          # no half-precision complex math
        else:
            real, imag = real.to(compute_dtype), imag.to(compute_dtype)
        support = self._shape[:-2]
        n = len(support)
        spec = torch.complex(real, imag) * math.sqrt(math.prod(support))
        kernel = torch.fft.irfftn(spec, s=support, dim=tuple(range(2, 2 + n))).movedim((0, 1), (-2, -1))
        return kernel if compute_dtype is None else kernel.to(compute_dtype)

    def get_config(self):
        return dict(initial_value=None, shape=tuple(map(int, self._shape)), dtype=str(self._dtype))


class GDNParameter(Parameter):
    """Non-negative parameterisation of the GDN parameters (parameters.py:186-269): the variable is
    sqrt(max(value + offset^2, offset^2)), the value lower_bound(variable, sqrt(minimum + offset^2))^2 - offset^2."""

    def __init__(self, initial_value, minimum=0.0, offset=2 ** -18, shape=None, dtype=None):
        super().__init__()
        self._minimum, self._offset = float(minimum), float(offset)
        if initial_value is None:
            if shape is None:
                raise ValueError("If initial_value is None, shape must be specified.")
            initial_value = torch.zeros(tuple(shape), dtype=dtype or torch.float32)
        else:
            initial_value = torch.as_tensor(initial_value, dtype=dtype)
        self.variable = torch.nn.Parameter(gdn_reparam_init(initial_value, self._offset))

    minimum = property(lambda self: self._minimum)
    offset = property(lambda self: self._offset)

    def forward(self, compute_dtype=None):
        variable = self.variable if compute_dtype is None else self.variable.to(compute_dtype)
        return gdn_reparam_value(variable, self._minimum, self._offset)

    def get_config(self):
        return dict(initial_value=None, minimum=self._minimum, offset=self._offset,
                    shape=tuple(self.variable.shape), dtype=str(self.variable.dtype))

