"""`SignalConv2D` (python/layers/signal_conv.py:61-1028), every 2-D configuration the reference implements: the
models' (`same_zeros`, explicit padding, one-sided square strides) straight on the fused kernels, the others (`valid`,
`same_reflect`, extra_pad_end=False, up + down strides, unequal strides, even supports, channel_separable) as a pad and a
crop around the same kernels."""
from __future__ import annotations

import math

import itertools
import os

import torch

from . import functional, parameters
from .gdn import GDN

__all__ = ["SignalConv2D"]


def _version_of(t):
    """The tensor's version counter, or None where it has none (tensors created under
    torch.inference_mode raise on `_version`)."""
    try:
        return t._version
    except RuntimeError:
        return None


def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else tuple(int(s) for s in v)


class SignalConv2D(torch.nn.Module):
    """Same constructor arguments as the reference (signal_conv.py:279-296).  Weights:
    `kernel_real`/`kernel_imag` (kernel_parameter="rdft") or `kernel`
    (kernel_parameter="variable"), and `bias` (signal_conv_test.py:38-40)."""

    def __init__(self, filters, kernel_support, corr=False, strides_down=1, strides_up=1,
                 padding="valid", extra_pad_end=True, channel_separable=False,
                 data_format="channels_last", activation=None, use_bias=False, use_explicit=True,
                 kernel_parameter="rdft", bias_parameter="variable", kernel_initializer=None,
                 bias_initializer=None, in_channels=None):
        super().__init__()
        self.filters = int(filters)
        self.kernel_support = _pair(kernel_support)
        self.corr = bool(corr)
        self.strides_down, self.strides_up = _pair(strides_down), _pair(strides_up)
        self.padding = str(padding).lower()
        if self.padding not in ("valid", "same_zeros", "same_reflect"):
            raise ValueError(f"Unsupported padding mode: '{padding}'.")
        self.extra_pad_end = bool(extra_pad_end)
        self.channel_separable = bool(channel_separable)
        self.data_format = data_format
        self.activation = activation
        self.use_bias = bool(use_bias)
        self.use_explicit = bool(use_explicit)
        # a tensor, a callable (e.g. a `parameters.Parameter`) or one of the strings (signal_conv.py:222-236)
        if isinstance(kernel_parameter, str) and kernel_parameter not in ("rdft", "variable"):
            raise ValueError("kernel_parameter must be a tensor, a callable, 'rdft' or 'variable'")
        if isinstance(bias_parameter, str) and bias_parameter != "variable":
            raise ValueError("bias_parameter must be a tensor, a callable or 'variable'")
        self.kernel_parameter = kernel_parameter if isinstance(kernel_parameter, str) else "given"
        self._kernel_given = None if isinstance(kernel_parameter, str) else kernel_parameter
        self._bias_given = None if isinstance(bias_parameter, str) else bias_parameter
        self._kernel_init, self._bias_init = kernel_initializer, bias_initializer
        self.kernel_real = self.kernel_imag = self.kernel_variable = self.bias = None
        self._check_implemented()
        if in_channels is not None:
            self.build(int(in_channels))

    def _raise_notimplemented(self):
        # (signal_conv.py:577-586: same text, so that callers' `assertRaisesRegex(NotImplementedError, "SignalConv")` hold)
        raise NotImplementedError(
            f"The provided combination of {type(self).__name__} arguments is not currently "
            f"implemented (filters={self.filters}, kernel_support={self.kernel_support}, "
            f"corr={self.corr}, strides_down={self.strides_down}, strides_up={self.strides_up}, "
            f"channel_separable={self.channel_separable}, data_format={self.data_format}, "
            f"padding={self.padding}). Try using odd-length kernels or turning off separability?")

    def _check_implemented(self):
        """The combinations the reference implements for rank 2 (signal_conv_test.py:317-349 `is_implemented`): anything
        else raises NotImplementedError, as there."""
        odd = all(s % 2 == 1 for s in self.kernel_support)
        upsampled = any(s != 1 for s in self.strides_up)
        can_use_transpose = not self.corr or odd
        must_use_transpose = upsampled or (not self.corr and not odd)
        if must_use_transpose and not can_use_transpose:
            self._raise_notimplemented()
        if self.channel_separable and (self.strides_up[0] != self.strides_up[1]
                                       or (must_use_transpose and self.filters != 1)):
            self._raise_notimplemented()

    def _is_model_configuration(self):
        """The configuration the models use and the fused paths serve directly: `same_zeros`, explicit padding, square
        strides on one side only, extra_pad_end."""
        return (self.padding == "same_zeros" and not self.channel_separable and self.use_explicit
                and self.extra_pad_end and self.strides_down[0] == self.strides_down[1]
                and self.strides_up[0] == self.strides_up[1]
                and (self.strides_down[0] == 1 or self.strides_up[0] == 1))

    def build(self, cin, device=None):
        if self.kernel_real is not None or self.kernel_variable is not None:
            return
        if self.use_bias and self._bias_given is None and self.bias is None:
            b = self._bias_init((self.filters,)) if self._bias_init else torch.zeros(self.filters)
            self.bias = torch.nn.Parameter(b.float().to(device))
        if self._kernel_given is not None:
            return
        kh, kw = self.kernel_support
        if self._kernel_init is not None:
            k = self._kernel_init((kh, kw, cin, self.filters))
        else:
            # Keras VarianceScaling(scale=1, fan_in, truncated normal) — signal_conv.py default
            std = math.sqrt(1.0 / (kh * kw * cin)) / 0.87962566103423978
            k = torch.empty(kh, kw, cin, self.filters)
            torch.nn.init.trunc_normal_(k, std=std, a=-2 * std, b=2 * std)
        k = k.float()
        if self.kernel_parameter == "rdft":
            real, imag = parameters.rdft_from_kernel(k)
            self.kernel_real = torch.nn.Parameter(real.to(device))
            self.kernel_imag = torch.nn.Parameter(imag.to(device))
        else:
            self.kernel_variable = torch.nn.Parameter(k.to(device))

    def _bias_value(self):
        """The bias in use: the layer's own variable `bias`, or the tensor / callable given as `bias_parameter`."""
        if not self.use_bias:
            return None
        if self._bias_given is not None:
            return torch.as_tensor(self._bias_given() if callable(self._bias_given) else self._bias_given)
        return self.bias

    @property
    def kernel(self):
        if self._kernel_given is not None:
            return torch.as_tensor(self._kernel_given() if callable(self._kernel_given) else self._kernel_given)
        if self.kernel_variable is not None:
            return self.kernel_variable
        if self.kernel_real is None:
            raise RuntimeError("Kernel is not initialized yet. Call build().")
        if torch.is_grad_enabled():
            return parameters.kernel_from_rdft(self.kernel_real, self.kernel_imag, self.kernel_support)
        # inference (compress / decompress run under no_grad): the inverse RDFT once per parameter version
        # instead of once per call — 11 small FFTs per bmshj2018 step, and an FFT plan shared by host
        # threads that code batch slices on different streams is not safe to execute concurrently
        key = (self.kernel_real.data_ptr(), _version_of(self.kernel_real), self.kernel_imag.data_ptr(),
               _version_of(self.kernel_imag), str(self.kernel_real.device))
        cached = getattr(self, "_kernel_cache", None)
        if key[1] is None or key[3] is None:
            cached = None                                     # inference tensors carry no version counter: recompute
        if cached is None or cached[0] != key:
            k = parameters.kernel_from_rdft(self.kernel_real, self.kernel_imag, self.kernel_support).contiguous()
            if k.is_cuda:
                torch.cuda.current_stream().synchronize()      # complete before another stream reads it
            object.__setattr__(self, "_kernel_cache", (key, k))
            cached = self._kernel_cache
        return cached[1]

    # One number per distinct value of a layer's weights (include/tfc_hip.h, tfc_conv2d_weights_key): the library keeps
    # the kernels' packed fragments of a keyed value between calls instead of packing them in front of every launch.
    _WEIGHT_KEYS = itertools.count(1)
    # Keyed (kept) packed weights under no_grad: on unless TFC_CONV_KEYED_WEIGHTS=0, or per layer / per class by
    # assigning `keyed_weights = False` (every call then packs its fragments from the tensor it is given, as training
    # does).  WHAT THE KEY SEES: the parameters' storage address, their autograd version counter and the layer's
    # `weights_generation`.  An in-place write through `.data` (`p.data.copy_(ema)`, manual weight loading) advances
    # neither address nor version — after such a write call `weights_changed()` (bumps the generation; the old
    # fragments are released in stream order) or `invalidate_kernel_cache()`.  load_state_dict, .to() / .cuda() /
    # .half(), train() / eval() do it themselves; optimizer steps and every other autograd-visible in-place op advance
    # the version counter.
    keyed_weights = os.environ.get("TFC_CONV_KEYED_WEIGHTS", "1") not in ("", "0")
    weights_generation = 0

    def weights_changed(self):
        """Tell the layer its weights were written behind autograd's back (`.data` writes): the kept inference kernel
        and the library's packed fragments of the old value are dropped."""
        self.weights_generation = self.weights_generation + 1
        self.invalidate_kernel_cache()

    def _inference_weights_key(self):
        """The key of the kernel's current value, or 0: gradients enabled (the weights are about to change), a kernel
        given as a tensor / callable (computed per call), parameters without version counters, or `keyed_weights` off."""
        if torch.is_grad_enabled() or self._kernel_given is not None or not self.keyed_weights:
            return 0
        src = (self.kernel_variable,) if self.kernel_variable is not None else (self.kernel_real, self.kernel_imag)
        if any(t is None or not t.is_cuda for t in src):
            return 0
        ident = tuple((t.data_ptr(), _version_of(t)) for t in src) + (str(src[0].device),)
        if any(v is None for _, v in ident[:-1]):
            return 0
        ident = ident + (self.weights_generation,)
        hit = self.__dict__.get("_wkey_cache")
        if hit is None or hit[0] != ident or hit[2] != id(self):
            if hit is not None and hit[2] == id(self):
                self._drop_weights_key(hit[1])
            hit = (ident, next(SignalConv2D._WEIGHT_KEYS), id(self))
            object.__setattr__(self, "_wkey_cache", hit)
        return hit[1]

    @staticmethod
    def _drop_weights_key(key):
        try:
            from .. import _lib
            lib = _lib.lib()
            lib.tfc_conv2d_drop_weights(key)
            lib.tfc_conv2d_drop_weights(key | (1 << 62))         # (of the flipped kernel: forward)
        except Exception:                                        # interpreter shutdown, library never loaded
            pass

    def __del__(self):
        hit = self.__dict__.get("_wkey_cache")
        if hit is not None and hit[2] == id(self):
            self._drop_weights_key(hit[1])

    def invalidate_kernel_cache(self):
        """Drops the cached inference kernel.  The cache is keyed on the parameters' storage and version
        counters, which in-place writes through `.data` (`p.data.copy_()`: EMA weight swaps, manual weight
        loading) do NOT advance — call this after such an update.  Loading a state dict, `.to()` / `.cuda()` /
        `.half()` and `train()` invalidate it themselves."""
        object.__setattr__(self, "_kernel_cache", None)
        hit = self.__dict__.get("_wkey_cache")
        if hit is not None:
            if hit[2] == id(self):
                self._drop_weights_key(hit[1])
            object.__setattr__(self, "_wkey_cache", None)

    def _load_from_state_dict(self, *args, **kwargs):
        self.invalidate_kernel_cache()
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self.invalidate_kernel_cache()
        return super()._apply(fn, *args, **kwargs)

    def train(self, mode=True):
        self.invalidate_kernel_cache()
        return super().train(mode)

    # ---------------------------------------------------------------------------------------------------------------
    # Every other configuration of the reference (`valid` — its default —, `same_reflect`, pre-padded `same_zeros`,
    # extra_pad_end=False, up- AND downsampling, unequal strides, even kernel supports, channel_separable), as a pad and a
    # crop around the same two kernels.  With u the zero-upsampled (pre-padded) input, the reference computes
    # (signal_conv.py:692-847)
    #   correlation:  c[i] = sum_t u[i + t] w[t]           ("valid"), kept at i = 0, sd, 2 sd, ...
    #   convolution:  f[m] = sum_j w[j] u[m - j]           ("full"),  kept at m = start, start + sd, ... < L_full - stop
    # and the kernels compute  corr_down_s(x)[i] = sum_t x[i s + t - k // 2] w[t]  (zeros outside x) and
    # conv_up_s(x)[n] = f[n + k // 2] over n in [0, len(x) s): a few zero samples in front / behind the input move the
    # kernels' windows onto the positions wanted, per dimension.
    # ---------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _zero_upsample(x, su, extra_pad_end):
        n, h, w, c = x.shape
        up = x.new_zeros((n, h * su[0], w * su[1], c))
        up[:, ::su[0], ::su[1]] = x
        return up if extra_pad_end else up[:, :h * su[0] - (su[0] - 1), :w * su[1] - (su[1] - 1)]

    def _forward_general(self, x, kernel):
        from ..ops.padding_ops import same_padding_for_kernel
        corr = self.corr
        ks, su, sd = self.kernel_support, self.strides_up, self.strides_down
        odd = all(s % 2 == 1 for s in ks)
        # the reference's kernel flips (signal_conv.py:861-880)
        if not corr and all(s == 1 for s in su) and odd:
            corr, kernel = True, kernel.flip(0, 1)
        elif corr and any(s != 1 for s in su) and odd:
            corr, kernel = False, kernel.flip(0, 1)
        if self.channel_separable:
            # out[..., c * F + f] = in[..., c] * kernel[..., c, f]: as a dense kernel that is zero off its diagonal blocks
            kh, kw, cin, f = kernel.shape
            dense = kernel.new_zeros((kh, kw, cin, cin * f))
            for ch in range(cin):
                dense[:, :, ch, ch * f:(ch + 1) * f] = kernel[:, :, ch]
            kernel = dense
        cin = x.shape[-1]
        if cin > 4 and cin % 16:
            # (the kernels take 1 ... 4 or a multiple of 16 input channels: zero channels change nothing)
            extra = 16 - cin % 16
            x = torch.nn.functional.pad(x, (0, extra))
            kernel = torch.nn.functional.pad(kernel, (0, 0, 0, extra))
        if self.padding == "valid":
            prepad = ((0, 0), (0, 0))
        else:
            prepad = same_padding_for_kernel(ks, corr, su)
            x = functional.pad2d(x, prepad[0], prepad[1], reflect=self.padding == "same_reflect")
        if corr and all(s == 1 for s in su):
            s = sd[0] if sd[0] == sd[1] else 1
            lens = [x.shape[1 + d] for d in range(2)]
            if any(lens[d] < ks[d] for d in range(2)):
                return x.new_zeros((x.shape[0], 0, 0, kernel.shape[-1]))
            e = [(-(ks[d] // 2)) % s for d in range(2)]
            xs = functional.pad2d(x, (e[0], 0), (e[1], 0))
            y = functional.conv2d_down(xs, kernel, None, s)
            sl = []
            for d in range(2):
                a = (ks[d] // 2 + e[d]) // s
                if s == sd[d]:
                    sl.append(slice(a, a + (lens[d] - ks[d]) // s + 1))
                else:
                    sl.append(slice(a, a + lens[d] - ks[d] + 1, sd[d]))
            return y[:, sl[0], sl[1]]
        if corr:
            self._raise_notimplemented()
        square = su[0] == su[1]
        s = su[0] if square else 1
        if not square:
            x = self._zero_upsample(x, su, True)
        pads, sl = [], []
        for d in range(2):
            k, length = ks[d], x.shape[1 + d]                  # length: of the kernel's input (upsampled already when not square)
            lup = length * s if square else length
            lfull = lup + (k - 1) - (0 if self.extra_pad_end else su[d] - 1)
            if self.padding == "valid":
                start = stop = k - 1
            else:
                start, stop = prepad[d][0] * su[d] + k // 2, prepad[d][1] * su[d] + (k - 1) // 2
            end = lfull - stop
            a = max(0, -(-(k // 2 - start) // s))
            b = max(0, -(-(end - k // 2 - lup) // s))
            pads.append((a, b))
            lo = start - k // 2 + a * s
            sl.append(slice(lo, max(lo, end - k // 2 + a * s), sd[d]))
        y = functional.conv2d_up(functional.pad2d(x, pads[0], pads[1]), kernel, None, s)
        return y[:, sl[0], sl[1]]

    def forward(self, inputs):
        if inputs.dim() != 4:
            raise ValueError(f"Input tensor must have rank 4, received shape {tuple(inputs.shape)}.")
        x = inputs.movedim(1, -1) if self.data_format == "channels_first" else inputs
        self.build(x.shape[-1], x.device)
        kernel = self.kernel
        if not self._is_model_configuration():
            y = self._forward_general(x, kernel.to(x.device))
            bias = self._bias_value()
            if bias is not None:
                y = y + bias.to(y.device, y.dtype)
            if self.activation is not None:
                y = torch.relu(y) if self.activation == "relu" else self.activation(y)
            return y.movedim(-1, 1) if self.data_format == "channels_first" else y
        act = self.activation
        fused = "relu" if act in (torch.relu, torch.nn.functional.relu, "relu") or isinstance(
            act, torch.nn.ReLU) else None
        corr, up, down = self.corr, self.strides_up[0], self.strides_down[0]
        odd = all(s % 2 == 1 for s in self.kernel_support)
        wkey = self._inference_weights_key() if x.is_cuda else 0
        if corr and up != 1:
            if not odd:
                self._check_implemented_fail()
            corr, kernel = False, kernel.flip(0, 1)            # signal_conv.py:875-880
            wkey = wkey | (1 << 62) if wkey else 0
        gdn = self._fusable_gdn(act, x, kernel, corr, up, down)
        if gdn is not None:
            # GDN / IGDN as the activation (signal_conv.py:948-950 applying gdn.py:371-421): one kernel where the
            # convolution kernel that takes the layer can (functional.conv2d_gdn), else the GDN kernel on its output
            prepared = act._prepared_params(gdn[0], gdn[1], x.dtype)
            y, done = functional.conv2d_gdn(x, kernel, self._bias_value(), down if corr else up, not corr, prepared,
                                            act.inverse, weights_key=wkey)
            if not done:
                y = functional.gdn_forward(y, gdn[0], gdn[1], act.inverse, False, 1.0, 1.0, prepared=prepared)
        else:
            if corr:
                y = functional.conv2d_down(x, kernel, self._bias_value(), down, fused, weights_key=wkey)
            else:
                y = functional.conv2d_up(x, kernel, self._bias_value(), up, fused, weights_key=wkey)
                if down != 1:
                    y = y[:, ::down, ::down]
            if act is not None and fused is None:
                y = act(y)
        return y.movedim(-1, 1) if self.data_format == "channels_first" else y

    # GDN / IGDN as the activation inside the convolution kernel (functional.conv2d_gdn): True / False, or None = by the
    # size of the layer's output (TFC_CONV_GDN=1 / 0 in the environment set it; unset = None).  Measured with steps in
    # flight (profiles/r04_notes.md): on bls2017 at 512 x 256x256 (outputs of 0.2 - 0.8 GB) the fused layers take
    # 9.5 -> 8.4 ms per step on one box, 8.78 -> 8.63 on another; on bmshj2018 at 128 x 768x512, whose 1.2 and 4.8 GB
    # maps are HBM-bound GDN launches of 0.55 and 2.2 ms, the step is level whether none, the small or all layers fuse
    # (four same-box comparisons), and a lone step wins 0.6 ms.  Default: no limit (TFC_CONV_GDN_MAX_MB = 0) — a model
    # step then has no GDN launches behind third-generation convolutions; a limit in MB restores the by-size rule.
    fuse_gdn_activation = {"": None, "0": False}.get(os.environ.get("TFC_CONV_GDN", ""), True)
    fuse_gdn_max_bytes = int(os.environ.get("TFC_CONV_GDN_MAX_MB", "0")) << 20
    # The image-side layer (three input channels) is different: its time is its output's HBM traffic, not the matrix
    # cores, and conv_image_gdn_kernel writes the normalised activations without the round trip (bmshj2018's first
    # layer at 128 x 768x512: 1.83 + 2.2 ms as two kernels).  On unless TFC_CONV_GDN_IMAGE=0.
    fuse_gdn_image = os.environ.get("TFC_CONV_GDN_IMAGE", "1") not in ("", "0")

    def _fusable_gdn(self, act, x, kernel, corr, up, down):
        """(beta, gamma) when `act` is a GDN layer in the configuration the fused entry point covers — inference on the
        layer's own variables, bfloat16, alpha = epsilon = 1, no rectification, channels-last inside — else None."""
        from .gdn import GDN
        image_side = self.fuse_gdn_image and corr and kernel.shape[-2] <= 4 and not getattr(act, "inverse", True)
        wanted = self.fuse_gdn_activation
        if wanted is None:
            scale = (1.0 / down if corr else float(up)) ** 2
            wanted = self.fuse_gdn_max_bytes <= 0 or \
                x.shape[0] * x.shape[1] * x.shape[2] * scale * kernel.shape[-1] * 2 <= self.fuse_gdn_max_bytes
        if not (wanted or image_side) or not isinstance(act, GDN) or torch.is_grad_enabled() \
                or not x.is_cuda or x.dtype != torch.bfloat16:
            return None
        if (not corr and down != 1) or act.rectify or act._beta_fixed is not None or act._gamma_fixed is not None:
            return None
        if act.data_format != "channels_last":          # (the activation is applied to the channels-last tensor inside)
            return None
        alpha, epsilon = act.alpha, act.epsilon
        if torch.is_tensor(alpha) or torch.is_tensor(epsilon) or float(alpha) != 1.0 or float(epsilon) != 1.0:
            return None
        cout = kernel.shape[-1]
        act.build(cout, x.device)
        beta, gamma = act.beta.to(x.device), act.gamma.to(x.device)
        if beta.shape != (cout,) or cout % 32 or cout > 256:
            return None
        return beta, gamma

    def _check_implemented_fail(self):
        raise NotImplementedError("cross-correlation with upsampling needs odd-length kernels")
