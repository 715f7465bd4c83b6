"""Fused training-time forward/backward of the entropy bottleneck with a deep factorized
prior (continuous_batched.py:291-322 + uniform_noise.py:117-156 + deep_factorized.py:166-194)
on the HIP kernels of csrc/factorized_bits.hip."""
from __future__ import annotations

import torch

from .. import _lib

__all__ = ["factorized_bits", "pack_factorized_params", "fused_factorized_supported", "fused_tail_mass"]

_DTYPE_CODE = {torch.float32: 0, torch.bfloat16: 1}
_BUILT = {(3, 3), (4, 3), (3, 5)}          # (layers, width) the library instantiates


def fused_factorized_supported(base, bottleneck, coding_rank) -> bool:
    """True if (prior, tensor) can take the fused path: equal hidden widths the library was
    built for, channels-last tensor on a HIP device whose coding unit is everything but the
    leading batch dimensions, one set of MLP weights per channel."""
    nf = getattr(base, "num_filters", None)
    if not nf or len(set(nf)) != 1 or (len(nf) + 1, nf[0]) not in _BUILT:
        return False
    if not bottleneck.is_cuda or bottleneck.dtype not in _DTYPE_CODE:
        return False
    bs = tuple(base.batch_shape)
    if len(bs) != 1 or bottleneck.dim() < 1 or bottleneck.shape[-1] != bs[0] or bs[0] > 512:
        return False
    return 1 <= coding_rank <= bottleneck.dim()


def pack_factorized_params(base) -> torch.Tensor:
    """[channels, P] float32 of the reparameterised MLP weights in the kernel's layout
    (differentiable: the chain rule through softplus / tanh stays with autograd)."""
    sp, th = torch.nn.functional.softplus, torch.tanh
    K = len(base.num_filters) + 1
    C = base.matrices[0].shape[0]
    parts = [sp(base.matrices[0]).reshape(C, -1), base.biases[0].reshape(C, -1), th(base.factors[0]).reshape(C, -1)]
    for l in range(1, K - 1):
        parts += [sp(base.matrices[l]).reshape(C, -1), base.biases[l].reshape(C, -1),
                  th(base.factors[l]).reshape(C, -1)]
    parts += [sp(base.matrices[K - 1]).reshape(C, -1), base.biases[K - 1].reshape(C, -1)]
    return torch.cat(parts, dim=1).to(torch.float32).contiguous()


class _FactorizedBits(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, noise, params, layers, width, units, elems, expected_grads=False, tail_mass=0.0):
        _lib.require_device()
        y = y.contiguous()
        y_hat = torch.empty_like(y)
        bits = torch.empty(units, dtype=torch.float32, device=y.device)
        channels = params.shape[0]
        noise_ptr = noise.data_ptr() if noise is not None else None
        if tail_mass:
            _lib.check(_lib.lib().tfc_factorized_bits_forward_tail(
                y.data_ptr(), noise_ptr, y_hat.data_ptr(), _DTYPE_CODE[y.dtype], units, elems, channels,
                params.data_ptr(), layers, width, tail_mass, None, bits.data_ptr(), _lib.stream_ptr()))
        else:
            _lib.check(_lib.lib().tfc_factorized_bits_forward(
                y.data_ptr(), noise_ptr, y_hat.data_ptr(), _DTYPE_CODE[y.dtype], units, elems, channels,
                params.data_ptr(), layers, width, None, bits.data_ptr(), _lib.stream_ptr()))
        if expected_grads:
            ctx.save_for_backward(y_hat, params, y)
        else:
            ctx.save_for_backward(y_hat, params)
        ctx.meta = (layers, width, units, elems, tail_mass)
        return y_hat, bits

    @staticmethod
    def backward(ctx, g_yhat, g_bits):
        y_hat, params = ctx.saved_tensors[:2]
        layers, width, units, elems, tail_mass = ctx.meta
        dy = torch.empty_like(y_hat)
        dparams = torch.zeros_like(params)
        gb = (g_bits if g_bits is not None else torch.zeros(units, device=y_hat.device)).to(torch.float32).contiguous()
        if tail_mass:
            # Laplace-mixture tail (continuous_base.py:298-334); the unperturbed input selects expected gradients
            y = ctx.saved_tensors[2] if len(ctx.saved_tensors) == 3 else None
            _lib.check(_lib.lib().tfc_factorized_bits_backward_tail(
                y.data_ptr() if y is not None else None, y_hat.data_ptr(), _DTYPE_CODE[y_hat.dtype], units, elems,
                params.shape[0], params.data_ptr(), layers, width, tail_mass, gb.data_ptr(), dy.data_ptr(),
                dparams.data_ptr(), _lib.stream_ptr()))
        elif len(ctx.saved_tensors) == 3:
            # expected gradients (math_ops.py:157-216): d/dy through the likelihood is the finite difference
            # of log p at y +- .5, evaluated at the unperturbed input
            y = ctx.saved_tensors[2]
            _lib.check(_lib.lib().tfc_factorized_bits_backward_expected(
                y.data_ptr(), y_hat.data_ptr(), _DTYPE_CODE[y_hat.dtype], units, elems, params.shape[0],
                params.data_ptr(), layers, width, gb.data_ptr(), dy.data_ptr(), dparams.data_ptr(),
                _lib.stream_ptr()))
        else:
            _lib.check(_lib.lib().tfc_factorized_bits_backward(
                y_hat.data_ptr(), _DTYPE_CODE[y_hat.dtype], units, elems, params.shape[0], params.data_ptr(),
                layers, width, gb.data_ptr(), dy.data_ptr(), dparams.data_ptr(), _lib.stream_ptr()))
        if g_yhat is not None:
            dy = dy + g_yhat
        return dy, None, dparams, None, None, None, None, None, None


def fused_tail_mass(laplace_tail_mass):
    """The entropy models' `laplace_tail_mass` as the kernels take it: 0.0 (none) or a float in (0, 1); None if
    the fused kernels cannot take it (a tensor — possibly learned — or out of range)."""
    if torch.is_tensor(laplace_tail_mass):
        return None
    m = float(laplace_tail_mass)
    return m if 0.0 <= m < 1.0 else None


def factorized_bits(bottleneck, base, coding_rank, noise=None, expected_grads=False, laplace_tail_mass=0.0):
    """(y_hat, bits): y_hat = bottleneck + noise (noise None: bottleneck itself), bits summed over
    the last `coding_rank` dimensions, shape = the leading dimensions.  expected_grads: the gradient of
    bits w.r.t. the bottleneck is the expectation over the noise (math_ops.py:157-216).  laplace_tail_mass:
    the likelihood is the mixture with a NoisyLaplace(0, 1) of continuous_base.py:298-334."""
    lead = bottleneck.shape[:bottleneck.dim() - coding_rank]
    units = 1
    for s in lead:
        units *= int(s)
    elems = bottleneck.numel() // max(units, 1)
    params = pack_factorized_params(base).to(bottleneck.device)
    if noise is not None:
        noise = noise.to(bottleneck.dtype).contiguous()
    y_hat, bits = _FactorizedBits.apply(bottleneck, noise, params, len(base.num_filters) + 1,
                                        int(base.num_filters[0]), units, elems, bool(expected_grads),
                                        float(laplace_tail_mass))
    return y_hat, bits.reshape(lead)


class _NoisyNormalBits(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, noise, scale, units, elems, expected_grads, tail_mass=0.0):
        _lib.require_device()
        y = y.contiguous()
        scale = scale.to(torch.float32).contiguous()
        y_hat = torch.empty_like(y)
        bits = torch.empty(units, dtype=torch.float32, device=y.device)
        noise_ptr = noise.data_ptr() if noise is not None else None
        if tail_mass:
            _lib.check(_lib.lib().tfc_noisy_normal_bits_forward_tail(
                y.data_ptr(), noise_ptr, scale.data_ptr(), y_hat.data_ptr(), _DTYPE_CODE[y.dtype], units, elems,
                tail_mass, bits.data_ptr(), _lib.stream_ptr()))
        else:
            _lib.check(_lib.lib().tfc_noisy_normal_bits_forward(
                y.data_ptr(), noise_ptr, scale.data_ptr(), y_hat.data_ptr(), _DTYPE_CODE[y.dtype], units, elems,
                bits.data_ptr(), _lib.stream_ptr()))
        ctx.save_for_backward(*((y_hat, scale, y) if expected_grads else (y_hat, scale)))
        ctx.meta = (units, elems, tail_mass)
        return y_hat, bits

    @staticmethod
    def backward(ctx, g_yhat, g_bits):
        y_hat, scale = ctx.saved_tensors[:2]
        y_in = ctx.saved_tensors[2] if len(ctx.saved_tensors) == 3 else None
        units, elems, tail_mass = ctx.meta
        dy = torch.empty_like(y_hat)
        dscale = torch.empty_like(scale)
        gb = (g_bits if g_bits is not None else torch.zeros(units, device=y_hat.device)).to(torch.float32).contiguous()
        y_in_ptr = y_in.data_ptr() if y_in is not None else None
        if tail_mass:
            _lib.check(_lib.lib().tfc_noisy_normal_bits_backward_tail(
                y_in_ptr, y_hat.data_ptr(), scale.data_ptr(), _DTYPE_CODE[y_hat.dtype], units, elems, tail_mass,
                gb.data_ptr(), dy.data_ptr(), dscale.data_ptr(), _lib.stream_ptr()))
        else:
            _lib.check(_lib.lib().tfc_noisy_normal_bits_backward(
                y_in_ptr, y_hat.data_ptr(), scale.data_ptr(), _DTYPE_CODE[y_hat.dtype], units, elems,
                gb.data_ptr(), dy.data_ptr(), dscale.data_ptr(), _lib.stream_ptr()))
        if g_yhat is not None:
            dy = dy + g_yhat
        return dy, None, dscale, None, None, None, None


def noisy_normal_bits(bottleneck, scale, coding_rank, noise=None, expected_grads=False, laplace_tail_mass=0.0):
    """(y_hat, bits) of a NoisyNormal(0, scale) prior, fused (csrc/noisy_normal_bits.hip): y_hat = bottleneck +
    noise, bits summed over the last `coding_rank` dimensions; `scale` broadcastable to the bottleneck's shape
    and differentiable (gradients flow to whatever produced it, e.g. the hyper-synthesis transform).
    laplace_tail_mass: mixture with a NoisyLaplace(0, 1) at the (shifted) bottleneck, continuous_base.py:298-334."""
    lead = bottleneck.shape[:bottleneck.dim() - coding_rank]
    units = 1
    for s in lead:
        units *= int(s)
    elems = bottleneck.numel() // max(units, 1)
    scale = torch.broadcast_to(scale.to(bottleneck.device), bottleneck.shape)
    if noise is not None:
        noise = noise.to(bottleneck.dtype).contiguous()
    y_hat, bits = _NoisyNormalBits.apply(bottleneck, noise, scale, units, elems, bool(expected_grads),
                                         float(laplace_tail_mass))
    return y_hat, bits.reshape(lead)


def fused_noisy_normal_supported(prior_fn, parameter_fns, bottleneck, coding_rank):
    from ..distributions import uniform_noise
    return (prior_fn is uniform_noise.NoisyNormal and set(parameter_fns) == {"loc", "scale"}
            and bottleneck.is_cuda and bottleneck.dtype in _DTYPE_CODE and coding_rank >= 1
            and bottleneck.numel() > 0)
