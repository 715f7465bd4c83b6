"""The reference's own SignalConv2D test (python/layers/signal_conv_test.py:171-349, 548-760): its 2-D cases, its SciPy
oracle (zero-insertion upsampling, scipy.signal correlate / convolve in `valid` mode, strided read-out) and its
`is_implemented` rule, restated for NHWC tensors.  Shared by the CPU tier (the layer's pad / crop arithmetic around
emulations of the two kernels) and the GPU tier (the layer on the kernels)."""
import itertools

import numpy as np


def numpy_upsample(x, strides_up, extra_pad_end):
    """x [N, C, H, W] (signal_conv_test.py:174-187)."""
    shape = np.array(x.shape, dtype=int)
    su = np.array(strides_up, dtype=int)
    shape[2:] *= su
    if not extra_pad_end:
        shape[2:] -= su - 1
    up = np.zeros(shape, dtype=np.float32)
    up[(slice(None), slice(None)) + tuple(slice(None, None, s) for s in su)] = x
    return up


def scipy_convolve_valid(corr, x, kernel, strides_down, strides_up, extra_pad_end, channel_separable):
    """x [N, C, H, W], kernel [kh, kw, Cin, F] -> [N, Cout, H', W'] (signal_conv_test.py:189-222)."""
    import scipy.signal
    base = scipy.signal.correlate if corr else scipy.signal.convolve
    convolve = lambda a, b, mode: base(a, b, mode=mode, method="direct")       # (exact sums of small integers; no FFT)
    slices = tuple(slice(None, None, s) for s in strides_down)
    if not all(s == 1 for s in strides_up):
        x = numpy_upsample(x, strides_up, extra_pad_end)
    cout = kernel.shape[-1] * (x.shape[1] if channel_separable else 1)
    probe = convolve(x[0, 0], kernel[..., 0, 0], mode="valid")[slices]
    out = np.zeros((x.shape[0], cout) + probe.shape, dtype=np.float64)
    for b in range(x.shape[0]):
        for f in range(kernel.shape[-1]):
            for c in range(x.shape[1]):
                co = c * kernel.shape[-1] + f if channel_separable else f
                out[b, co] += convolve(x[b, c].astype(np.float64), kernel[..., c, f].astype(np.float64), mode="valid")[slices]
    return out


def is_implemented(kernel_support, corr, strides_up, channel_separable, filters):
    """signal_conv_test.py:317-349 for rank 2."""
    odd = all(s % 2 == 1 for s in kernel_support)
    can_use_transpose = not corr or odd
    must_use_transpose = any(s != 1 for s in strides_up) or (not corr and not odd)
    if must_use_transpose and not can_use_transpose:
        return False
    if channel_separable and any(s != strides_up[0] for s in strides_up):
        return False
    if channel_separable and must_use_transpose and filters != 1:
        return False
    return True


def valid_cases():
    """test_2d_valid_spatial + test_2d_valid_channels + test_2d_bias_activation."""
    for sep in (False, True):
        for support in ((10, 7), (5, 8)):
            for ks in ((5, 2), (2, 3), (3, 3)):
                for corr in (False, True):
                    for sd, su, epe in zip([(1, 1), (2, 2), (1, 1), (1, 1), (3, 5)], [(1, 1), (1, 1), (2, 2), (4, 3), (1, 1)],
                                           [True, True, False, True, True]):
                        yield dict(input_support=support, channels=1, filters=1, kernel_support=ks, corr=corr,
                                   strides_down=sd, strides_up=su, extra_pad_end=epe, channel_separable=sep, use_bias=False)
    for sep in (False, True):
        for channels, filters in zip([1, 2], [2, 1]):
            for su in ((1, 1), (2, 2)):
                yield dict(input_support=(8, 7), channels=channels, filters=filters, kernel_support=(3, 3), corr=False,
                           strides_down=(1, 1), strides_up=su, extra_pad_end=False, channel_separable=sep, use_bias=False)
    yield dict(input_support=(4, 6), channels=1, filters=1, kernel_support=(2, 2), corr=True, strides_down=(1, 1),
               strides_up=(1, 1), extra_pad_end=True, channel_separable=False, use_bias=True)
    # beyond the reference's list: the default-argument layer at a model's width, several channels with strides
    yield dict(input_support=(9, 11), channels=5, filters=3, kernel_support=(3, 3), corr=True, strides_down=(2, 2),
               strides_up=(1, 1), extra_pad_end=True, channel_separable=False, use_bias=True)
    yield dict(input_support=(7, 6), channels=16, filters=4, kernel_support=(5, 5), corr=False, strides_down=(1, 1),
               strides_up=(2, 2), extra_pad_end=True, channel_separable=False, use_bias=False)
    yield dict(input_support=(6, 7), channels=3, filters=2, kernel_support=(3, 3), corr=True, strides_down=(2, 2),
               strides_up=(2, 2), extra_pad_end=True, channel_separable=False, use_bias=False)


def same_cases():
    """test_2d_same_zeros_spatial (explicit and pre-padded) + test_2d_same_padding: identity kernels."""
    for support in ((4, 7), (5, 6)):
        for ks in ((3, 2), (2, 6), (3, 3)):
            for corr in (False, True):
                for sd, su, epe in zip([(1, 1), (1, 1), (1, 1), (3, 5), (2, 3)], [(1, 1), (2, 3), (5, 2), (1, 1), (3, 2)],
                                       [True, False, True, True, False]):
                    for explicit in (True, False):
                        yield dict(input_support=support, kernel_support=ks, corr=corr, strides_down=sd, strides_up=su,
                                   extra_pad_end=epe, padding="same_zeros", use_explicit=explicit)
    yield dict(input_support=(4, 5), kernel_support=(3, 2), corr=True, strides_down=(1, 1), strides_up=(1, 1),
               extra_pad_end=True, padding="same_reflect", use_explicit=True)
    yield dict(input_support=(6, 7), kernel_support=(3, 3), corr=False, strides_down=(1, 1), strides_up=(2, 2),
               extra_pad_end=True, padding="same_reflect", use_explicit=True)


def identity_kernel(kernel_support, corr):
    """initializers.IdentityInitializer (python/layers/initializers.py:25-63) for one channel and one filter: a unit
    impulse at the kernel's centre — support // 2 (the layer's convention for both corr and convolution there)."""
    k = np.zeros(tuple(kernel_support) + (1, 1), np.float32)
    k[kernel_support[0] // 2, kernel_support[1] // 2, 0, 0] = 1.0
    return k


def same_reflect_oracle(x, kernel, kernel_support, corr):
    """`same_reflect` with a general kernel, strides 1: reflect pre-pad (padding_ops.same_padding_for_kernel) and a
    `valid` correlation / convolution.  x [N, C, H, W]."""
    (t, b), (l, r) = [(s // 2, (s - 1) // 2) if corr else ((s - 1) // 2, s // 2) for s in kernel_support]
    xp = np.pad(x, ((0, 0), (0, 0), (t, b), (l, r)), mode="reflect")
    return scipy_convolve_valid(corr, xp, kernel, (1, 1), (1, 1), True, False)
