"""Thin torch-tensor front-ends of the transform kernels in libtfc_hip.so."""
from __future__ import annotations

import torch

from .. import _lib

_DTYPE_CODE = {torch.float32: 0, torch.bfloat16: 1}


class GDNPrepared:
    """The forward kernels' LDS image of one (beta, gamma) pair, built once (include/tfc_hip.h
    tfc_gdn_params_create): the per-call preparation launch drops out of inference calls."""

    def __init__(self, beta: torch.Tensor, gamma: torch.Tensor, dtype: torch.dtype):
        device = _lib.require_device()
        if dtype not in _DTYPE_CODE:
            raise TypeError(f"GDN kernel supports float32 and bfloat16, got {dtype}")
        beta = beta.detach().to(device, torch.float32).contiguous()
        gamma = gamma.detach().to(device, torch.float32).contiguous()
        C = beta.shape[0]
        if gamma.shape != (C, C):
            raise ValueError(f"beta/gamma shapes {tuple(beta.shape)}/{tuple(gamma.shape)} do not match")
        import ctypes
        out = ctypes.c_void_p()
        _lib.check(_lib.lib().tfc_gdn_params_create(beta.data_ptr(), gamma.data_ptr(), C, _DTYPE_CODE[dtype],
                                                    _lib.stream_ptr(), ctypes.byref(out)))
        self.ptr, self.channels, self.dtype = out, C, dtype
        self._keep = (beta, gamma)        # read by the preparation kernel in stream order

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                _lib.lib().tfc_gdn_params_destroy(self.ptr)
        except Exception:
            pass


def gdn_forward(x: torch.Tensor, beta: torch.Tensor, gamma: torch.Tensor, inverse: bool = False,
                rectify: bool = False, alpha: float = 1, epsilon: float = 1, prepared: GDNPrepared = None) -> torch.Tensor:
    """Fused GDN/IGDN forward, channels-last: x [..., C], beta [C], gamma [C(in), C(out)]
    (the layout of `GDN.gamma`, python/layers/gdn.py:394-398).  `prepared` (GDNPrepared of the same beta /
    gamma / dtype): skips the per-call parameter preparation (fixed alpha in {1, 2}, epsilon in {1, .5})."""
    _lib.require_device()
    if x.dtype not in _DTYPE_CODE:
        raise TypeError(f"GDN kernel supports float32 and bfloat16, got {x.dtype}")
    alpha, epsilon = float(alpha), float(epsilon)
    x = x.contiguous()
    C = x.shape[-1]
    if prepared is not None and alpha in (1, 2) and epsilon in (1, 0.5):
        if prepared.channels != C or prepared.dtype != x.dtype:
            raise ValueError("prepared GDN parameters do not match the input")
        y = torch.empty_like(x)
        _lib.check(_lib.lib().tfc_gdn_forward_prepared(
            prepared.ptr, x.data_ptr(), y.data_ptr(), x.numel() // C, int(bool(inverse)), int(bool(rectify)),
            int(alpha), 1 if epsilon == 0.5 else 0, _lib.stream_ptr()))
        return y
    beta = beta.detach().to(x.device, torch.float32).contiguous()
    gamma = gamma.detach().to(x.device, torch.float32).contiguous()
    if beta.shape != (C,) or gamma.shape != (C, C):
        raise ValueError(f"beta/gamma shapes {tuple(beta.shape)}/{tuple(gamma.shape)} do not match C={C}")
    y = torch.empty_like(x)
    if alpha not in (1, 2) or epsilon not in (1, 0.5):
        # general exponents (gdn.py:386-387, :411-412: tf.pow), the kernels' GEN variant
        _lib.check(_lib.lib().tfc_gdn_forward_general(
            x.data_ptr(), y.data_ptr(), _DTYPE_CODE[x.dtype], x.numel() // C, C, beta.data_ptr(),
            gamma.data_ptr(), int(bool(inverse)), int(bool(rectify)), alpha, epsilon, _lib.stream_ptr()))
        return y
    _lib.check(_lib.lib().tfc_gdn_forward(
        x.data_ptr(), y.data_ptr(), _DTYPE_CODE[x.dtype], x.numel() // C, C, beta.data_ptr(),
        gamma.data_ptr(), int(bool(inverse)), int(bool(rectify)), int(alpha),
        1 if epsilon == 0.5 else 0, _lib.stream_ptr()))
    return y


def _conv(fn_name, x, kernel, bias, stride, activation, up, weights_key=0):
    _lib.require_device()
    if x.dtype not in _DTYPE_CODE:
        raise TypeError(f"conv kernel supports float32 and bfloat16, got {x.dtype}")
    if x.dim() != 4:
        raise ValueError(f"Input tensor must have rank 4, received shape {tuple(x.shape)}.")
    x = x.contiguous()
    n, h, w, cin = x.shape
    kh, kw, kcin, cout = kernel.shape
    if kcin != cin:
        raise ValueError(f"kernel expects {kcin} input channels, input has {cin}")
    kernel = kernel.detach().to(x.device, torch.float32).contiguous()
    if bias is not None:
        bias = bias.detach().to(x.device, torch.float32).contiguous()
    if up:
        oh, ow = h * stride, w * stride
    else:
        oh, ow = -(-h // stride), -(-w // stride)
    y = torch.empty((n, oh, ow, cout), dtype=x.dtype, device=x.device)
    act = {None: 0, "relu": 1}[activation]
    if weights_key:
        _lib.lib().tfc_conv2d_weights_key(weights_key)      # this thread's next conv call: fragments packed once per value
    _lib.check(getattr(_lib.lib(), fn_name)(
        x.data_ptr(), kernel.data_ptr(), None if bias is None else bias.data_ptr(), y.data_ptr(),
        _DTYPE_CODE[x.dtype], n, h, w, cin, cout, kh, kw, int(stride), act, _lib.stream_ptr()))
    return y


def conv2d_gdn(x, kernel, bias, stride, up, prepared: GDNPrepared, inverse: bool, weights_key=0):
    """SignalConv2D with GDN / IGDN as its activation (inference, bfloat16): -> (y, fused).  fused: the convolution
    kernel applied the activation itself (include/tfc_hip.h, tfc_conv2d_gdn); else y is the convolution's output and
    the caller applies the GDN kernel."""
    import ctypes
    _lib.require_device()
    x = x.contiguous()
    n, h, w, cin = x.shape
    kh, kw, kcin, cout = kernel.shape
    if kcin != cin:
        raise ValueError(f"kernel expects {kcin} input channels, input has {cin}")
    kernel = kernel.detach().to(x.device, torch.float32).contiguous()
    if bias is not None:
        bias = bias.detach().to(x.device, torch.float32).contiguous()
    oh, ow = (h * stride, w * stride) if up else (-(-h // stride), -(-w // stride))
    y = torch.empty((n, oh, ow, cout), dtype=x.dtype, device=x.device)
    fused = ctypes.c_int(0)
    if weights_key:
        _lib.lib().tfc_conv2d_weights_key(weights_key)
    _lib.check(_lib.lib().tfc_conv2d_gdn(
        x.data_ptr(), kernel.data_ptr(), None if bias is None else bias.data_ptr(), y.data_ptr(),
        _DTYPE_CODE[x.dtype], n, h, w, cin, cout, kh, kw, int(stride), int(bool(up)), prepared.ptr, int(bool(inverse)),
        ctypes.byref(fused), _lib.stream_ptr()))
    return y, bool(fused.value)


def conv2d_wgrad(a, b, kernel_support, stride, transpose):
    """Weight gradient kernel: G[t][ca][cb] = sum A[n, q*s + t - k/2, ca] B[n, q, cb] as a float32
    [kh, kw, Cin, Cout] tensor (transpose=True: A carries Cout, B carries Cin)."""
    _lib.require_device()
    kh, kw = kernel_support
    ca, cb = a.shape[-1], b.shape[-1]
    built = (256, 192, 128, 64, 32)
    if any(c > 4 and c not in built for c in (ca, cb)) and ca % 32 == 0 and cb % 32 == 0:
        # the kernel is built for 32, 64, 128, 192 or 256 channels on either side; the gradient of a channel
        # block pair only needs those channels, so other widths (ms2020: 224, 320 .. 512) go in blocks
        def blocks(c):
            if c <= 4 or c in built:
                return [(0, c)]
            out, i = [], 0
            while i < c:
                w = next(w for w in built if w <= c - i)
                out.append((i, i + w))
                i += w
            return out
        rows = []
        for a0, a1 in blocks(ca):
            cols = [conv2d_wgrad(a[..., a0:a1], b[..., b0:b1], kernel_support, stride, transpose)
                    for b0, b1 in blocks(cb)]
            rows.append(torch.cat(cols, dim=2 if transpose else 3))
        return torch.cat(rows, dim=3 if transpose else 2)
    a, b = a.contiguous(), b.contiguous()
    n, ha, wa, ca = a.shape
    _, hb, wb, cb = b.shape
    shape = (kh, kw, cb, ca) if transpose else (kh, kw, ca, cb)
    dw = torch.zeros(shape, dtype=torch.float32, device=a.device)
    _lib.check(_lib.lib().tfc_conv2d_wgrad(
        a.data_ptr(), b.data_ptr(), dw.data_ptr(), _DTYPE_CODE[a.dtype], n, ha, wa, ca, hb, wb, cb,
        kh, kw, int(stride), int(bool(transpose)), _lib.stream_ptr()))
    return dw


class _ConvFunction(torch.autograd.Function):
    """Differentiable wrapper of the two conv entry points (the reference differentiates
    signal_conv.py:663-690 / 778-847 with TF autodiff):
      dx  = the OTHER direction's forward kernel on dy with the kernel's channel axes swapped,
      dw  = tfc_conv2d_wgrad,   dbias = sum of dy over pixels."""

    @staticmethod
    def forward(ctx, x, kernel, bias, stride, activation, up):
        y = _conv("tfc_conv2d_up" if up else "tfc_conv2d_down", x, kernel, bias, stride, activation, up)
        ctx.save_for_backward(x, kernel, y if activation == "relu" else None)
        ctx.meta = (stride, activation, up, bias is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, kernel, y = ctx.saved_tensors
        stride, activation, up, has_bias = ctx.meta
        gy = gy.to(x.dtype).contiguous()
        if activation == "relu":
            gy = gy * (y > 0)
        kh, kw = kernel.shape[:2]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            kt = kernel.permute(0, 1, 3, 2)
            if up:
                dx = _conv("tfc_conv2d_down", gy, kt, None, stride, None, False)
            else:
                dx = _conv("tfc_conv2d_up", gy, kt, None, stride, None, True)[:, :x.shape[1], :x.shape[2]]
        if ctx.needs_input_grad[1]:
            dw = conv2d_wgrad(gy, x, (kh, kw), stride, True) if up else conv2d_wgrad(x, gy, (kh, kw), stride, False)
            dw = dw.to(kernel.dtype)
        if has_bias and ctx.needs_input_grad[2]:
            db = gy.float().sum(dim=(0, 1, 2))
        return dx, dw, db, None, None, None


def _conv_dispatch(x, kernel, bias, stride, activation, up, weights_key=0):
    needs = torch.is_grad_enabled() and (x.requires_grad or kernel.requires_grad
                                         or (bias is not None and bias.requires_grad))
    if needs:
        return _ConvFunction.apply(x, kernel, bias, stride, activation, up)
    return _conv("tfc_conv2d_up" if up else "tfc_conv2d_down", x, kernel, bias, stride, activation, up, weights_key)


def conv2d_down(x, kernel, bias=None, stride=1, activation=None, weights_key=0):
    """Analysis correlation (signal_conv.py:663-690): NHWC x, HWIO kernel, `same_zeros`.  weights_key: a number that
    names this VALUE of `kernel` (include/tfc_hip.h, tfc_conv2d_weights_key): its packed fragments are kept between
    calls; 0: packed per call."""
    return _conv_dispatch(x, kernel, bias, stride, activation, False, weights_key)


def conv2d_up(x, kernel, bias=None, stride=1, activation=None, weights_key=0):
    """Synthesis transposed convolution (signal_conv.py:778-847, extra_pad_end=True)."""
    return _conv_dispatch(x, kernel, bias, stride, activation, True, weights_key)


def gdn_backward(x, grad, beta, gamma, inverse=False, rectify=False, alpha=1, epsilon=1):
    """Gradients of gdn_forward w.r.t. (x, beta, gamma) on the HIP kernel (alpha in {1, 2}, epsilon in {1, .5};
    the general exponents go through `gdn_general_composite`)."""
    _lib.require_device()
    if alpha not in (1, 2) or epsilon not in (1, 0.5):
        raise NotImplementedError("tfc_gdn_backward implements alpha in {1, 2} and epsilon in {1, .5}")
    x = x.contiguous()
    grad = grad.contiguous()
    C = x.shape[-1]
    beta = beta.detach().to(x.device, torch.float32).contiguous()
    gamma = gamma.detach().to(x.device, torch.float32).contiguous()
    dx = torch.empty_like(x)
    dbeta = torch.zeros(C, dtype=torch.float32, device=x.device)
    dgamma = torch.zeros(C, C, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().tfc_gdn_backward(
        x.data_ptr(), grad.data_ptr(), dx.data_ptr(), _DTYPE_CODE[x.dtype], x.numel() // C, C,
        beta.data_ptr(), gamma.data_ptr(), int(bool(inverse)), int(bool(rectify)), int(alpha),
        1 if epsilon == 0.5 else 0, dbeta.data_ptr(), dgamma.data_ptr(), _lib.stream_ptr()))
    return dx, dbeta, dgamma


def gdn_general_composite(x, beta, gamma, alpha, epsilon, inverse=False, rectify=False):
    """The layer with general exponents as differentiable device tensor ops (float32 arithmetic): the
    training path of learned alpha / epsilon, whose gradients d/dalpha and d/depsilon the fused backward
    kernel does not produce.  Same formula and op order as gdn.py:377-416; the forward-only (inference)
    path of the same configuration is `gdn_forward` on the HIP kernel."""
    _lib.require_device()
    xf = x.float()
    if rectify:
        xf = torch.relu(xf)
    norm = torch.pow(xf, alpha) @ gamma.to(xf.device, torch.float32) + beta.to(xf.device, torch.float32)
    norm = torch.pow(norm, epsilon)
    y = xf * norm if inverse else xf / norm
    return y.to(x.dtype)


def image_to_unit(x, dtype):
    """uint8 image tensor -> dtype(x) / 255 in one pass (the input scaling of the models' analysis transforms:
    bls2017.py:164-170, bmshj2018.py:219-224).  Other inputs take the torch expression."""
    if x.dtype != torch.uint8 or dtype not in _DTYPE_CODE or not x.is_cuda or not x.is_contiguous():
        return x.to(dtype) / 255.0
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    _lib.check(_lib.lib().tfc_image_to_unit(x.data_ptr(), y.data_ptr(), _DTYPE_CODE[dtype], x.numel(), _lib.stream_ptr()))
    return y


def unit_to_image(x):
    """saturate_cast(round(x * 255), uint8) with the product rounded to x's dtype, in one pass (the output of the
    models' synthesis transforms: bls2017.py:186-190, bmshj2018.py:262-264)."""
    if x.dtype not in _DTYPE_CODE or not x.is_cuda or not x.is_contiguous():
        return torch.clamp(torch.round((x * 255.0).float()), 0, 255).to(torch.uint8)
    y = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    _lib.check(_lib.lib().tfc_unit_to_image(x.data_ptr(), _DTYPE_CODE[x.dtype], y.data_ptr(), x.numel(), _lib.stream_ptr()))
    return y


def index_prepare(indexes, num_tables):
    """int32(min(max(indexes, 0), num_tables - 1)) in one pass, or None where the torch ops have to do it (integer or
    non-contiguous indexes, gradients wanted)."""
    if (indexes.dtype not in _DTYPE_CODE or not indexes.is_cuda or not indexes.is_contiguous()
            or (torch.is_grad_enabled() and indexes.requires_grad)):
        return None
    out = torch.empty(indexes.shape, dtype=torch.int32, device=indexes.device)
    _lib.check(_lib.lib().tfc_index_prepare(indexes.data_ptr(), _DTYPE_CODE[indexes.dtype], out.data_ptr(),
                                            indexes.numel(), int(num_tables), _lib.stream_ptr()))
    return out


def pad2d(x, pad_h, pad_w, reflect=False):
    """Spatial padding of an NHWC tensor: zeros (tf.pad CONSTANT) or mirror without the edge sample (tf.pad REFLECT) —
    SignalConv2D's pre-pad (signal_conv.py:880-893).  One HBM-bound kernel (tfc_pad2d) where no gradient is wanted and
    the tensor is on the device; differentiable tensor ops (zero pad / index gathers) otherwise."""
    (t, b), (l, r) = (int(pad_h[0]), int(pad_h[1])), (int(pad_w[0]), int(pad_w[1]))
    if t == b == l == r == 0:
        return x
    n, h, w, c = x.shape
    if reflect and (max(t, b) >= h or max(l, r) >= w):
        raise ValueError(f"reflect padding {(t, b), (l, r)} must be smaller than the input's {(h, w)}")
    needs = torch.is_grad_enabled() and x.requires_grad
    if x.is_cuda and not needs and x.element_size() in (2, 4):
        x = x.contiguous()
        y = torch.empty((n, h + t + b, w + l + r, c), dtype=x.dtype, device=x.device)
        _lib.check(_lib.lib().tfc_pad2d(x.data_ptr(), y.data_ptr(), x.element_size(), n, h, w, c, t, b, l, r,
                                        int(bool(reflect)), _lib.stream_ptr()))
        return y
    if not reflect:
        return torch.nn.functional.pad(x, (0, 0, l, r, t, b))

    def mirror(length, before, after):
        i = torch.arange(-before, length + after, device=x.device).abs()
        return torch.where(i >= length, 2 * (length - 1) - i, i)
    return x.index_select(1, mirror(h, t, b)).index_select(2, mirror(w, l, r))
