"""Soft-rounding layers (python/layers/soft_round.py:27-62)."""
import torch

from ..ops import round_ops

__all__ = ["SoftRound", "SoftRoundConditionalMean"]


class SoftRound(torch.nn.Module):
    """Differentiable approximation of rounding (or its inverse)."""

    def __init__(self, alpha=5.0, inverse=False):
        super().__init__()
        self._alpha = alpha
        self._transform = round_ops.soft_round_inverse if inverse else round_ops.soft_round

    def forward(self, inputs):
        return self._transform(inputs, self._alpha)


class SoftRoundConditionalMean(torch.nn.Module):
    """Conditional mean of the inputs given noisy soft-rounded values."""

    def __init__(self, alpha=5.0):
        super().__init__()
        self._alpha = alpha

    def forward(self, inputs):
        return round_ops.soft_round_conditional_mean(inputs, alpha=self._alpha)
