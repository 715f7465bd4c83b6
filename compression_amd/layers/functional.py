"""Thin torch-tensor front-ends of the transform kernels in libtfc_hip.so."""
from __future__ import annotations

import torch

from .. import _lib

_DTYPE_CODE = {torch.float32: 0, torch.bfloat16: 1}


def gdn_forward(x: torch.Tensor, beta: torch.Tensor, gamma: torch.Tensor, inverse: bool = False,
                rectify: bool = False, alpha: float = 1, epsilon: float = 1) -> torch.Tensor:
    """Fused GDN/IGDN forward, channels-last: x [..., C], beta [C], gamma [C(in), C(out)]
    (the layout of `GDN.gamma`, python/layers/gdn.py:394-398)."""
    _lib.require_device()
    if x.dtype not in _DTYPE_CODE:
        raise TypeError(f"GDN kernel supports float32 and bfloat16, got {x.dtype}")
    if alpha not in (1, 2) or epsilon not in (1, 0.5):
        raise NotImplementedError("GDN kernel implements alpha in {1, 2} and epsilon in {1, .5}")
    x = x.contiguous()
    C = x.shape[-1]
    beta = beta.detach().to(x.device, torch.float32).contiguous()
    gamma = gamma.detach().to(x.device, torch.float32).contiguous()
    if beta.shape != (C,) or gamma.shape != (C, C):
        raise ValueError(f"beta/gamma shapes {tuple(beta.shape)}/{tuple(gamma.shape)} do not match C={C}")
    y = torch.empty_like(x)
    _lib.check(_lib.lib().tfc_gdn_forward(
        x.data_ptr(), y.data_ptr(), _DTYPE_CODE[x.dtype], x.numel() // C, C, beta.data_ptr(),
        gamma.data_ptr(), int(bool(inverse)), int(bool(rectify)), int(alpha),
        1 if epsilon == 0.5 else 0, _lib.stream_ptr()))
    return y
