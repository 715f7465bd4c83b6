"""Rounding helpers (python/ops/round_ops.py:28-133): straight-through round, and the soft rounding of
"Universally Quantized Neural Compression" (Agustsson & Theis), Sec. 4.1."""
import torch

__all__ = ["round_st", "soft_round", "soft_round_inverse", "soft_round_conditional_mean"]


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


def _alpha_tensor(alpha, like):
    return torch.as_tensor(alpha, dtype=like.dtype, device=like.device)


def soft_round(x, alpha, eps=1e-3):
    """Differentiable approximation of round (round_ops.py:46-75): identity as alpha -> 0 (exactly, below
    `eps`), round as alpha -> inf.  m = floor(x) + 1/2, y = m + tanh(alpha (x - m)) / (2 tanh(alpha / 2))."""
    x = torch.as_tensor(x)
    alpha = _alpha_tensor(alpha, x)
    bounded = torch.clamp(alpha, min=eps)        # keeps the unused branch of the where() free of NaN gradients
    m = torch.floor(x) + 0.5
    y = m + torch.tanh(bounded * (x - m)) / (torch.tanh(bounded / 2.0) * 2.0)
    return torch.where(alpha < eps, x, y)


def soft_round_inverse(y, alpha, eps=1e-3):
    """Inverse of `soft_round` (round_ops.py:78-108); the result is kept inside its half-integer cell even
    where atanh overflows."""
    y = torch.as_tensor(y)
    alpha = _alpha_tensor(alpha, y)
    bounded = torch.clamp(alpha, min=eps)
    m = torch.floor(y) + 0.5
    r = torch.atanh((y - m) * (torch.tanh(bounded / 2.0) * 2.0)) / bounded
    r = torch.clamp(r, -0.5, 0.5)
    return torch.where(alpha < eps, y, m + r)


def soft_round_conditional_mean(y, alpha):
    """E[Y | soft_round(Y) + U = y] for U ~ U(-1/2, 1/2) and Y locally uniform (round_ops.py:111-133)."""
    return soft_round_inverse(torch.as_tensor(y) - 0.5, alpha) + 0.5

