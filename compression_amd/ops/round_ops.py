"""Rounding helpers (python/ops/round_ops.py:28-42)."""
import torch

__all__ = ["round_st"]


class _RoundST(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, offset):
        if offset is None:
            return torch.round(inputs)
        return torch.round(inputs - offset) + offset

    @staticmethod
    def backward(ctx, grad):
        return grad, None


def round_st(inputs, offset=None):
    """Straight-through round (half-to-even like tf.round) with optional offset."""
    return _RoundST.apply(inputs, offset)
