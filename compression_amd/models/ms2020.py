"""Minnen & Singh 2020 channel-wise autoregressive model (models/ms2020.py:40-430): hyperprior giving latent
means and scales, the main latent cut into `num_slices` channel slices that are coded in order, each with
(mu, sigma) predicted from the hyperprior features and the slices decoded so far, plus a latent residual
prediction per slice.  The slice loop is strictly sequential on the decoder side (a slice's parameters need the
previous slices' reconstructions): it is the latency-critical user of the indexed coder ops."""
from __future__ import annotations

import math

import numpy as np
import torch

from .. import distributions, entropy_models, layers

__all__ = ["AnalysisTransform", "SynthesisTransform", "HyperAnalysisTransform", "HyperSynthesisTransform",
           "SliceTransform", "MS2020Model"]


def _conv(C, k, cin, **kw):
    return layers.SignalConv2D(C, (k, k), padding="same_zeros", in_channels=cin, **kw)


class AnalysisTransform(torch.nn.Module):
    """ms2020.py:53-69."""

    def __init__(self, latent_depth, num_filters=192):
        super().__init__()
        C = num_filters
        kw = dict(corr=True, strides_down=2, use_bias=True)
        self.layer_0 = _conv(C, 5, 3, activation=layers.GDN(), **kw)
        self.layer_1 = _conv(C, 5, C, activation=layers.GDN(), **kw)
        self.layer_2 = _conv(C, 5, C, activation=layers.GDN(), **kw)
        self.layer_3 = _conv(latent_depth, 5, C, activation=None, **kw)

    def forward(self, x):
        return self.layer_3(self.layer_2(self.layer_1(self.layer_0(x / 255.0))))


class SynthesisTransform(torch.nn.Module):
    """ms2020.py:72-92."""

    def __init__(self, latent_depth, num_filters=192):
        super().__init__()
        C = num_filters
        kw = dict(corr=False, strides_up=2, use_bias=True)
        g = lambda: layers.GDN(inverse=True)
        self.layer_0 = _conv(C, 5, latent_depth, activation=g(), **kw)
        self.layer_1 = _conv(C, 5, C, activation=g(), **kw)
        self.layer_2 = _conv(C, 5, C, activation=g(), **kw)
        self.layer_3 = _conv(3, 5, C, activation=None, **kw)

    def forward(self, y):
        return self.layer_3(self.layer_2(self.layer_1(self.layer_0(y)))) * 255.0


class HyperAnalysisTransform(torch.nn.Module):
    """ms2020.py:95-113."""

    def __init__(self, latent_depth, hyperprior_depth):
        super().__init__()
        kw = dict(corr=True)
        self.layer_0 = _conv(320, 3, latent_depth, strides_down=1, use_bias=True, activation="relu", **kw)
        self.layer_1 = _conv(256, 5, 320, strides_down=2, use_bias=True, activation="relu", **kw)
        self.layer_2 = _conv(hyperprior_depth, 5, 256, strides_down=2, use_bias=False, activation=None, **kw)

    def forward(self, y):
        return self.layer_2(self.layer_1(self.layer_0(y)))


class HyperSynthesisTransform(torch.nn.Module):
    """ms2020.py:116-138: the output is still latent (ReLU at the end on purpose)."""

    def __init__(self, hyperprior_depth):
        super().__init__()
        kw = dict(corr=False, use_bias=True, kernel_parameter="variable", activation="relu")
        self.layer_0 = _conv(192, 5, hyperprior_depth, strides_up=2, **kw)
        self.layer_1 = _conv(256, 5, 192, strides_up=2, **kw)
        self.layer_2 = _conv(320, 3, 256, strides_up=1, **kw)

    def forward(self, z):
        return self.layer_2(self.layer_1(self.layer_0(z)))


class SliceTransform(torch.nn.Module):
    """ms2020.py:141-167: channel-conditional parameters / latent residual prediction of one slice."""

    def __init__(self, in_channels, slice_depth):
        super().__init__()
        kw = dict(corr=False, strides_up=1, use_bias=True, kernel_parameter="variable")
        self.layer_0 = _conv(224, 5, in_channels, activation="relu", **kw)
        self.layer_1 = _conv(128, 5, 224, activation="relu", **kw)
        self.layer_2 = _conv(slice_depth, 3, 128, activation=None, **kw)

    def forward(self, t):
        return self.layer_2(self.layer_1(self.layer_0(t)))


class MS2020Model(torch.nn.Module):
    """ms2020.py:170-430.  compress() returns (x_shape, y_shape, z_shape, z_string, *y_strings) like the
    reference; strings are per image (the reference's tf.function takes one image; here a batch)."""

    def __init__(self, lmbda=0.01, num_filters=192, latent_depth=320, hyperprior_depth=192, num_slices=10,
                 max_support_slices=5, num_scales=64, scale_min=0.11, scale_max=256.0,
                 compute_dtype=torch.float32):
        super().__init__()
        if latent_depth % num_slices:
            raise ValueError("Slices do not evenly divide latent depth (%d / %d)" % (latent_depth, num_slices))
        self.lmbda, self.num_scales, self.num_slices = lmbda, num_scales, num_slices
        self.max_support_slices = max_support_slices
        self.latent_depth, self.compute_dtype = latent_depth, compute_dtype
        offset = math.log(scale_min)
        factor = (math.log(scale_max) - math.log(scale_min)) / (num_scales - 1.0)
        self.scale_fn = lambda i: torch.exp(offset + factor * i)
        self.analysis_transform = AnalysisTransform(latent_depth, num_filters)
        self.synthesis_transform = SynthesisTransform(latent_depth, num_filters)
        self.hyper_analysis_transform = HyperAnalysisTransform(latent_depth, hyperprior_depth)
        self.hyper_synthesis_mean_transform = HyperSynthesisTransform(hyperprior_depth)
        self.hyper_synthesis_scale_transform = HyperSynthesisTransform(hyperprior_depth)
        sd = latent_depth // num_slices
        support = lambda k: 320 + sd * (k if max_support_slices < 0 else min(k, max_support_slices))
        self.cc_mean_transforms = torch.nn.ModuleList(SliceTransform(support(k), sd) for k in range(num_slices))
        self.cc_scale_transforms = torch.nn.ModuleList(SliceTransform(support(k), sd) for k in range(num_slices))
        self.lrp_transforms = torch.nn.ModuleList(SliceTransform(support(k) + sd, sd) for k in range(num_slices))
        self.hyperprior = distributions.NoisyDeepFactorized(batch_shape=(hyperprior_depth,))
        self.em_y = self.em_z = None

    @property
    def container_dtypes(self):
        """.tfci layout = decompress()'s signature (ms2020.py:390, :560): three shapes, then the strings."""
        return [np.int32] * 3 + [bytes] * (1 + self.num_slices)

    def _models(self, compression):
        em_z = entropy_models.ContinuousBatchedEntropyModel(
            self.hyperprior, coding_rank=3, compression=compression, offset_heuristic=False,
            bottleneck_dtype=self.compute_dtype)
        em_y = entropy_models.LocationScaleIndexedEntropyModel(
            distributions.NoisyNormal, self.num_scales, self.scale_fn, coding_rank=3, compression=compression,
            bottleneck_dtype=self.compute_dtype)
        return em_y, em_z

    def init_compression(self):
        self.em_y, self.em_z = self._models(True)
        return self

    def _hyper_features(self, z_hat, y_shape):
        """Latent scale / mean features, cropped to the latent's extent: for image sizes that are multiples of
        64 (what the reference supports: its concat of these features with a decoded slice needs equal
        extents) the crop is the identity."""
        ls = self.hyper_synthesis_scale_transform(z_hat)[:, :y_shape[0], :y_shape[1], :]
        lm = self.hyper_synthesis_mean_transform(z_hat)[:, :y_shape[0], :y_shape[1], :]
        return ls.contiguous(), lm.contiguous()

    def _slice_params(self, k, latent_means, latent_scales, y_hat_slices, y_shape):
        support = y_hat_slices if self.max_support_slices < 0 else y_hat_slices[:self.max_support_slices]
        mean_support = torch.cat([latent_means] + support, dim=-1)
        mu = self.cc_mean_transforms[k](mean_support)[:, :y_shape[0], :y_shape[1], :]
        scale_support = torch.cat([latent_scales] + support, dim=-1)
        sigma = self.cc_scale_transforms[k](scale_support)[:, :y_shape[0], :y_shape[1], :]
        return mean_support, mu, sigma

    def _lrp(self, k, mean_support, y_hat_slice):
        lrp = self.lrp_transforms[k](torch.cat([mean_support, y_hat_slice], dim=-1))
        return y_hat_slice + 0.5 * torch.tanh(lrp)

    def forward(self, x, training=True):
        """(loss, bpp, mse) — ms2020.py:200-262."""
        em_y, em_z = self._models(False)
        x = x.to(self.compute_dtype)
        y = self.analysis_transform(x)
        y_shape = tuple(y.shape[1:-1])
        z = self.hyper_analysis_transform(y)
        num_pixels = x.shape[1] * x.shape[2]
        _, z_bits = em_z(z, training=training)
        bpp = z_bits.mean() / num_pixels
        z_hat = em_z.quantize(z).to(self.compute_dtype)
        latent_scales, latent_means = self._hyper_features(z_hat, y_shape)
        y_hat_slices = []
        for k, y_slice in enumerate(torch.chunk(y, self.num_slices, dim=-1)):
            mean_support, mu, sigma = self._slice_params(k, latent_means, latent_scales, y_hat_slices, y_shape)
            _, slice_bits = em_y(y_slice, sigma, loc=mu, training=training)
            bpp = bpp + slice_bits.mean() / num_pixels
            y_hat_slice = em_y.quantize(y_slice, loc=mu).to(self.compute_dtype)
            y_hat_slices.append(self._lrp(k, mean_support, y_hat_slice))
        x_hat = self.synthesis_transform(torch.cat(y_hat_slices, dim=-1))
        x_hat = x_hat[:, :x.shape[1], :x.shape[2], :]
        mse = torch.mean((x.float() - x_hat.float()) ** 2).to(bpp.dtype)
        return bpp + self.lmbda * mse, bpp, mse

    @torch.no_grad()
    def compress(self, x):
        """uint8 [B, H, W, 3] -> (x_shape, y_shape, z_shape, z_string[B], y_string_0[B], ...) — ms2020.py:334-382."""
        if x.dim() == 3:
            x = x[None]
        x = x.to(self.compute_dtype)
        y = self.analysis_transform(x)
        z = self.hyper_analysis_transform(y)
        x_shape, y_shape, z_shape = tuple(x.shape[1:-1]), tuple(y.shape[1:-1]), tuple(z.shape[1:-1])
        z_string = self.em_z.compress(z)
        z_hat = self.em_z.decompress(z_string, z_shape).to(self.compute_dtype)
        latent_scales, latent_means = self._hyper_features(z_hat, y_shape)
        residuals, scales, y_hat_slices = [], [], []
        for k, y_slice in enumerate(torch.chunk(y, self.num_slices, dim=-1)):
            mean_support, mu, sigma = self._slice_params(k, latent_means, latent_scales, y_hat_slices, y_shape)
            # what the coder gets for this slice (ms2020.py:362-364: compress(y_slice, sigma, loc=mu) codes y_slice - mu).
            # The encoder needs no slice's string to go on — only the decoder's chain is serial — so the slices'
            # coder calls go out together behind the loop, ONE launch per stage for all of them.
            residuals.append((y_slice - mu).contiguous())
            scales.append(sigma.contiguous())
            # What the decoder will see.  The reference decodes the string it has just written
            # (ms2020.py:366); decode(encode(y)) is round(y - mu) + mu, which quantize() computes without the
            # slice's serial decode (half of compress()'s coder time); the round-trip test checks that the two
            # sides reconstruct the same image bit for bit.
            y_hat_slice = self.em_y.quantize(y_slice, loc=mu).to(self.compute_dtype)
            y_hat_slices.append(self._lrp(k, mean_support, y_hat_slice))
        from ..ops import gen_ops
        y_strings = [gen_ops.fetch_strings(h) for h in self.em_y.compress_many(residuals, scales)]
        return (x_shape, y_shape, z_shape, z_string) + tuple(y_strings)

    @torch.no_grad()
    def decompress(self, x_shape, y_shape, z_shape, z_string, *y_strings):
        """ms2020.py:384-420: slice k can only be decoded after slices < k."""
        assert len(y_strings) == self.num_slices
        z_hat = self.em_z.decompress(z_string, z_shape).to(self.compute_dtype)
        latent_scales, latent_means = self._hyper_features(z_hat, y_shape)
        y_hat_slices, checks = [], []
        for k, s in enumerate(y_strings):
            mean_support, mu, sigma = self._slice_params(k, latent_means, latent_scales, y_hat_slices, y_shape)
            # (the decoder's sanity flags stay on the device until the last slice is enqueued: a read-back per slice
            # would put a host round trip between every two links of the slice chain)
            y_hat_slice, ok = self.em_y.decompress(s, sigma.contiguous(), mu.contiguous(), defer_sanity=True)
            checks.append(ok)
            y_hat_slices.append(self._lrp(k, mean_support, y_hat_slice.to(self.compute_dtype)))
        x_hat = self.synthesis_transform(torch.cat(y_hat_slices, dim=-1))[:, :x_shape[0], :x_shape[1], :]
        out = torch.clamp(torch.round(x_hat.float()), 0, 255).to(torch.uint8)
        if self.em_y.decode_sanity_check and not bool(torch.stack([c.all() for c in checks]).all()):
            raise RuntimeError("Sanity check failed.")
        return out


if __name__ == "__main__":      # python -m compression_amd.models.ms2020 compress in.png out.tfci
    import sys

    from .codec_io import main
    sys.exit(main(MS2020Model))
