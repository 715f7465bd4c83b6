"""Reparameterisations of layer weights (python/layers/parameters.py:69-269).
Host-side float math on the weights, once per forward; no HIP kernel."""
from __future__ import annotations

import math

import torch

from ..ops import math_ops

__all__ = ["rdft_from_kernel", "kernel_from_rdft", "gdn_reparam_init", "gdn_reparam_value"]


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
