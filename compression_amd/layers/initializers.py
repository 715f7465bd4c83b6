"""Initializers for layer classes (python/layers/initializers.py:25-63)."""
import torch

__all__ = ["IdentityInitializer"]


class IdentityInitializer:
    """An n-D convolution kernel [*support, in, out] whose output reproduces its input (except possibly at
    the boundaries): `gain` at the support's centre tap on the channel diagonal."""

    def __init__(self, gain=1):
        self.gain = gain

    def __call__(self, shape, dtype=None):
        shape = tuple(int(s) for s in shape)
        if len(shape) <= 2:
            raise ValueError(f"shape must be at least rank 3, got {shape}.")
        dtype = dtype or torch.float32
        kernel = torch.zeros(shape, dtype=dtype)
        centre = tuple(s // 2 for s in shape[:-2])
        kernel[centre] = self.gain * torch.eye(shape[-2], shape[-1], dtype=dtype)
        return kernel

    def get_config(self):
        return dict(gain=self.gain)
