"""CPU tier: SignalConv2D's pad / crop arithmetic for every 2-D configuration of the reference's own test
(python/layers/signal_conv_test.py: `valid`, pre-padded `same_zeros`, `same_reflect`, extra_pad_end, up + down strides,
unequal strides, even supports, channel_separable) against that test's SciPy oracle.  The two HIP kernels are replaced
here by torch-CPU statements of what they compute (pinned to SciPy on the GPU by
tests/test_signal_conv_gpu.py):
    corr_down_s(x)[i] = sum_t x[i s + t - k // 2] w[t]              i < ceil(len / s), zeros outside x
    conv_up_s(x)[n]   = sum_j w[j] u[n + k // 2 - j]                n < len s, u = x with s - 1 zeros behind every sample
so that the layer's host logic — the only thing that is new around the kernels — is checked without a device; the GPU
tier runs the same cases on the kernels themselves."""
import numpy as np
import pytest
import torch

import signal_conv_cases as cases


def emu_down(x, kernel, bias=None, stride=1, activation=None, weights_key=0):
    kh, kw = kernel.shape[:2]
    n, h, w, c = x.shape
    assert c <= 4 or c % 16 == 0, "the kernels take 1..4 or a multiple of 16 input channels"
    oh, ow = -(-h // stride), -(-w // stride)
    xp = torch.nn.functional.pad(x.permute(0, 3, 1, 2).double(), (kw // 2, kw + stride, kh // 2, kh + stride))
    y = torch.nn.functional.conv2d(xp, kernel.permute(3, 2, 0, 1).double(), stride=stride)[:, :, :oh, :ow]
    return y.permute(0, 2, 3, 1).to(x.dtype).contiguous()


def emu_up(x, kernel, bias=None, stride=1, activation=None, weights_key=0):
    kh, kw = kernel.shape[:2]
    n, h, w, c = x.shape
    assert c <= 4 or c % 16 == 0
    # conv_transpose2d = the full convolution f of the zero-upsampled input: f[m] = sum_q x[q] w[m - q s]
    f = torch.nn.functional.conv_transpose2d(x.permute(0, 3, 1, 2).double(), kernel.permute(2, 3, 0, 1).double(), stride=stride)
    f = torch.nn.functional.pad(f, (0, stride + kw, 0, stride + kh))
    y = f[:, :, kh // 2:kh // 2 + h * stride, kw // 2:kw // 2 + w * stride]
    return y.permute(0, 2, 3, 1).to(x.dtype).contiguous()


@pytest.fixture
def emulated(monkeypatch):
    from compression_amd.layers import functional
    monkeypatch.setattr(functional, "conv2d_down", emu_down)
    monkeypatch.setattr(functional, "conv2d_up", emu_up)


def run_layer(kernel, x_nchw, **kw):
    from compression_amd import layers
    filters = kernel.shape[-1]
    layer = layers.SignalConv2D(filters, kw.pop("kernel_support"), kernel_parameter=torch.from_numpy(kernel), **kw)
    with torch.no_grad():
        y = layer(torch.from_numpy(np.moveaxis(x_nchw, 1, -1).copy()))
    return np.moveaxis(y.numpy(), -1, 1), layer


def test_emulations_state_the_same_zeros_kernels():
    """The emulations against the GPU tier's own `same_zeros` oracle (tests/test_signal_conv_gpu.py scipy_same_zeros)."""
    from test_signal_conv_gpu import scipy_same_zeros
    rng = np.random.default_rng(0)
    for kshape, stride, up, shape in (((5, 5, 3, 2), 2, False, (1, 7, 9, 3)), ((3, 3, 2, 3), 1, False, (2, 5, 6, 2)),
                                      ((5, 5, 2, 3), 2, True, (1, 4, 5, 2)), ((9, 9, 1, 2), 4, True, (1, 3, 4, 1)),
                                      ((3, 3, 1, 1), 1, True, (1, 6, 5, 1))):
        x = rng.integers(0, 8, shape).astype(np.float32)
        k = rng.integers(-3, 4, kshape).astype(np.float32)
        fn = emu_up if up else emu_down
        got = fn(torch.from_numpy(x), torch.from_numpy(k), None, stride).numpy()
        assert np.allclose(got, scipy_same_zeros(x, k, stride, up), atol=1e-9), (kshape, stride, up)      # (SciPy may take its FFT path)


@pytest.mark.parametrize("case", list(cases.valid_cases()), ids=lambda c: "-".join(f"{k[:2]}{v}" for k, v in c.items()))
def test_valid_against_scipy(emulated, case):
    case = dict(case)
    rng = np.random.default_rng(1)
    support, channels, filters = case.pop("input_support"), case.pop("channels"), case.pop("filters")
    x = rng.integers(0, 32, (1, channels) + support).astype(np.float32)
    kernel = rng.integers(0, 16, case["kernel_support"] + (channels, filters)).astype(np.float32)
    if not cases.is_implemented(case["kernel_support"], case["corr"], case["strides_up"], case["channel_separable"], filters):
        from compression_amd import layers
        with pytest.raises(NotImplementedError, match="SignalConv"):
            layers.SignalConv2D(filters, case["kernel_support"], corr=case["corr"], strides_down=case["strides_down"],
                                strides_up=case["strides_up"], channel_separable=case["channel_separable"],
                                kernel_parameter=torch.from_numpy(kernel))
        return
    want = cases.scipy_convolve_valid(case["corr"], x, kernel, case["strides_down"], case["strides_up"],
                                      case["extra_pad_end"], case["channel_separable"])
    got, layer = run_layer(kernel, x, padding="valid", activation=(lambda t: t) if case["use_bias"] else None, **case)
    assert got.shape == want.shape
    assert np.array_equal(got, want)            # small integers: every sum exact


@pytest.mark.parametrize("case", list(cases.same_cases()), ids=lambda c: "-".join(f"{k[:2]}{v}" for k, v in c.items()))
def test_same_identity_kernels(emulated, case):
    """signal_conv_test.py:262-315 `run_same`: with the identity kernel the layer returns its input, up- and downsampled."""
    case = dict(case)
    support = case.pop("input_support")
    x = np.arange(np.prod(support), dtype=np.float32).reshape((1, 1) + support)
    if not cases.is_implemented(case["kernel_support"], case["corr"], case["strides_up"], False, 1):
        from compression_amd import layers
        with pytest.raises(NotImplementedError, match="SignalConv"):
            layers.SignalConv2D(1, case["kernel_support"], corr=case["corr"], strides_up=case["strides_up"],
                                strides_down=case["strides_down"], padding=case["padding"])
        return
    kernel = cases.identity_kernel(case["kernel_support"], case["corr"])
    got, _ = run_layer(kernel, x, **case)
    want = x
    if not all(s == 1 for s in case["strides_up"]):
        want = cases.numpy_upsample(want, case["strides_up"], case["extra_pad_end"])
    want = want[(slice(None), slice(None)) + tuple(slice(None, None, s) for s in case["strides_down"])]
    assert got.shape == want.shape
    assert np.array_equal(got, want)


@pytest.mark.parametrize("corr", [True, False])
@pytest.mark.parametrize("ks", [(3, 3), (5, 3), (4, 3)])
def test_same_reflect_general_kernel(emulated, corr, ks):
    if not cases.is_implemented(ks, corr, (1, 1), False, 2):
        pytest.skip("not implemented by the reference either")
    rng = np.random.default_rng(2)
    x = rng.integers(0, 32, (2, 3, 7, 9)).astype(np.float32)
    kernel = rng.integers(0, 16, ks + (3, 2)).astype(np.float32)
    got, _ = run_layer(kernel, x, kernel_support=ks, corr=corr, padding="same_reflect")
    want = cases.same_reflect_oracle(x, kernel, ks, corr)
    assert got.shape == want.shape == (2, 2, 7, 9)
    assert np.array_equal(got, want)


def test_default_arguments_construct_and_run(emulated):
    """`SignalConv2D(filters, k)` — padding="valid", convolution, rdft kernel — is the reference's default layer."""
    from compression_amd import layers
    torch.manual_seed(0)
    layer = layers.SignalConv2D(4, 3)
    x = torch.randn(2, 8, 9, 3)
    with torch.no_grad():
        y = layer(x)
    assert tuple(y.shape) == (2, 6, 7, 4)
    want = cases.scipy_convolve_valid(False, np.moveaxis(x.numpy(), -1, 1), layer.kernel.detach().numpy(), (1, 1), (1, 1), True, False)
    assert np.allclose(np.moveaxis(y.numpy(), -1, 1), want, atol=1e-5)
