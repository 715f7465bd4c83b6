"""Tail / offset helpers for range-coding tables
(python/distributions/helpers.py:29-219)."""
from __future__ import annotations

import math

import torch

__all__ = ["estimate_tails", "quantization_offset", "lower_tail", "upper_tail"]


def estimate_tails(func, target, shape, dtype, device=None):
    """Solves func(x) == target element-wise for monotonic `func` with the reference's
    simple Adam-like iteration (helpers.py:29-104): halve-averaged moments, step
    0.1 / sqrt(count+1), 100 more iterations after the gradient first flips sign;
    returns the best iterate seen."""
    shape = tuple(int(s) for s in shape)
    target = torch.as_tensor(target, dtype=dtype, device=device)
    tails = torch.zeros(shape, dtype=dtype, device=device)
    m = torch.zeros_like(tails)
    v = torch.ones_like(tails)
    count = torch.zeros(shape, dtype=torch.int32, device=device)
    best_tails = tails.clone()
    best_loss = torch.full(shape, torch.finfo(dtype).max, dtype=dtype, device=device)
    loss = best_loss.clone()
    while bool(loss.max() > 1e-8) and bool(count.min() < 100):
        x = tails.detach().requires_grad_(True)
        with torch.enable_grad():
            loss_t = torch.abs(func(x) - target)
            grad, = torch.autograd.grad(loss_t.sum(), x)
        loss = loss_t.detach()
        better = loss < best_loss
        best_tails = torch.where(better, tails, best_tails)
        best_loss = torch.where(better, loss, best_loss)
        prev_m = m
        m = (m + grad) / 2
        v = (v + grad * grad) / 2
        k = torch.sqrt((count + 1).to(dtype))
        tails = tails - 0.1 * m / (k * torch.sqrt(v) + 1e-20)
        count = torch.where((count > 0) | (prev_m * grad < 0), count + 1, count)
    return best_tails


def quantization_offset(distribution):
    """mode / median / mean modulo 1 in [-.5, .5] (helpers.py:107-147)."""
    offset = None
    for name in ("_quantization_offset", "mode", None, "mean"):
        try:
            if name is None:
                offset = distribution.quantile(0.5)
            else:
                offset = getattr(distribution, name)()
            break
        except (AttributeError, NotImplementedError):
            continue
    if offset is None:
        offset = torch.zeros((), dtype=distribution.dtype)
    offset = offset.detach()
    return offset - torch.round(offset)


def lower_tail(distribution, tail_mass):
    """Cut-off such that ~tail_mass/2 lies below (helpers.py:150-183)."""
    try:
        tail = distribution._lower_tail(tail_mass)
    except (AttributeError, NotImplementedError):
        try:
            tail = distribution.quantile(tail_mass / 2)
        except NotImplementedError:
            try:
                tail = estimate_tails(distribution.log_cdf, math.log(tail_mass / 2),
                                      distribution.batch_shape, distribution.dtype)
            except NotImplementedError:
                raise NotImplementedError(
                    "`distribution` must implement `_lower_tail()`, `quantile()`, or "
                    "`log_cdf()` so that lower tail can be located.")
    return tail.detach()


def upper_tail(distribution, tail_mass):
    """Cut-off such that ~tail_mass/2 lies above (helpers.py:186-219)."""
    try:
        tail = distribution._upper_tail(tail_mass)
    except (AttributeError, NotImplementedError):
        try:
            tail = distribution.quantile(1 - tail_mass / 2)
        except NotImplementedError:
            try:
                tail = estimate_tails(distribution.log_survival_function, math.log(tail_mass / 2),
                                      distribution.batch_shape, distribution.dtype)
            except NotImplementedError:
                raise NotImplementedError(
                    "`distribution` must implement `_upper_tail()`, `quantile()`, or "
                    "`log_survival_function()` so that upper tail can be located.")
    return tail.detach()
