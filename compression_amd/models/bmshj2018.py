"""Ballé, Minnen, Singh, Hwang, Johnston 2018 scale-hyperprior model
(models/bmshj2018.py:53-264)."""
from __future__ import annotations

import math

import torch

from .. import distributions, entropy_models, layers
from ..layers import functional
from ..pipeline import inline_lane

__all__ = ["AnalysisTransform", "SynthesisTransform", "HyperAnalysisTransform",
           "HyperSynthesisTransform", "BMSHJ2018Model"]


def _conv(C, k, cin, **kw):
    return layers.SignalConv2D(C, (k, k), padding="same_zeros", in_channels=cin, **kw)


class AnalysisTransform(torch.nn.Module):
    def __init__(self, num_filters):
        super().__init__()
        C = num_filters
        self.layer_0 = _conv(C, 5, 3, corr=True, strides_down=2, use_bias=True, activation=layers.GDN())
        self.layer_1 = _conv(C, 5, C, corr=True, strides_down=2, use_bias=True, activation=layers.GDN())
        self.layer_2 = _conv(C, 5, C, corr=True, strides_down=2, use_bias=True, activation=layers.GDN())
        self.layer_3 = _conv(C, 5, C, corr=True, strides_down=2, use_bias=True, activation=None)

    def forward(self, x):
        return self.unit(x / 255.0)

    def unit(self, u):
        """The layers, on an image already scaled to [0, 1] (`functional.image_to_unit`)."""
        return self.layer_3(self.layer_2(self.layer_1(self.layer_0(u))))


class SynthesisTransform(torch.nn.Module):
    def __init__(self, num_filters):
        super().__init__()
        C = num_filters
        g = lambda: layers.GDN(inverse=True)
        self.layer_0 = _conv(C, 5, C, corr=False, strides_up=2, use_bias=True, activation=g())
        self.layer_1 = _conv(C, 5, C, corr=False, strides_up=2, use_bias=True, activation=g())
        self.layer_2 = _conv(C, 5, C, corr=False, strides_up=2, use_bias=True, activation=g())
        self.layer_3 = _conv(3, 5, C, corr=False, strides_up=2, use_bias=True, activation=None)

    def forward(self, y):
        return self.unit(y) * 255.0

    def unit(self, y):
        """The layers, without the scaling to [0, 255] (`functional.unit_to_image` takes it with the rounding)."""
        return self.layer_3(self.layer_2(self.layer_1(self.layer_0(y))))


class HyperAnalysisTransform(torch.nn.Module):
    def __init__(self, num_filters):
        super().__init__()
        C = num_filters
        self.layer_0 = _conv(C, 3, C, corr=True, strides_down=1, use_bias=True, activation="relu")
        self.layer_1 = _conv(C, 5, C, corr=True, strides_down=2, use_bias=True, activation="relu")
        self.layer_2 = _conv(C, 5, C, corr=True, strides_down=2, use_bias=False, activation=None)

    def forward(self, y):
        return self.layer_2(self.layer_1(self.layer_0(y)))


class HyperSynthesisTransform(torch.nn.Module):
    def __init__(self, num_filters):
        super().__init__()
        C = num_filters
        kw = dict(corr=False, use_bias=True, kernel_parameter="variable")
        self.layer_0 = _conv(C, 5, C, strides_up=2, activation="relu", **kw)
        self.layer_1 = _conv(C, 5, C, strides_up=2, activation="relu", **kw)
        self.layer_2 = _conv(C, 3, C, strides_up=1, activation=None, **kw)

    def forward(self, z):
        return self.layer_2(self.layer_1(self.layer_0(z)))


class BMSHJ2018Model(torch.nn.Module):
    # layout of the .tfci container: [string, side_string, x_shape, y_shape, z_shape]
    # (bmshj2018.py:219-240, 355-358)
    num_strings, num_packed = 2, 5

    def __init__(self, lmbda=0.01, num_filters=192, num_scales=64, scale_min=0.11, scale_max=256.0,
                 compute_dtype=torch.float32):
        super().__init__()
        self.lmbda, self.num_scales = lmbda, num_scales
        self.compute_dtype = compute_dtype
        offset = math.log(scale_min)
        factor = (math.log(scale_max) - math.log(scale_min)) / (num_scales - 1.0)
        self.scale_fn = lambda i: torch.exp(offset + factor * i)
        self.analysis_transform = AnalysisTransform(num_filters)
        self.synthesis_transform = SynthesisTransform(num_filters)
        self.hyper_analysis_transform = HyperAnalysisTransform(num_filters)
        self.hyper_synthesis_transform = HyperSynthesisTransform(num_filters)
        self.hyperprior = distributions.NoisyDeepFactorized(batch_shape=(num_filters,))
        self.entropy_model = self.side_entropy_model = None

    def _models(self, compression):
        em = entropy_models.LocationScaleIndexedEntropyModel(
            distributions.NoisyNormal, self.num_scales, self.scale_fn, coding_rank=3,
            compression=compression, bottleneck_dtype=self.compute_dtype)
        side = entropy_models.ContinuousBatchedEntropyModel(
            self.hyperprior, coding_rank=3, compression=compression, bottleneck_dtype=self.compute_dtype)
        return em, side

    def forward(self, x, training=True):
        em, side = self._models(False)
        x = x.to(self.compute_dtype)
        y = self.analysis_transform(x)
        z = self.hyper_analysis_transform(torch.abs(y))
        z_hat, side_bits = side(z, training=training)
        indexes = self.hyper_synthesis_transform(z_hat.to(self.compute_dtype))
        y_hat, bits = em(y, indexes, training=training)
        x_hat = self.synthesis_transform(y_hat.to(self.compute_dtype))
        num_pixels = x.shape[0] * x.shape[1] * x.shape[2]
        bpp = (bits.sum() + side_bits.sum()) / num_pixels
        mse = torch.mean((x.float() - x_hat.float()) ** 2).to(bpp.dtype)
        return bpp + self.lmbda * mse, bpp, mse

    def init_compression(self):
        self.entropy_model, self.side_entropy_model = self._models(True)
        return self

    @torch.no_grad()
    def compress(self, x, device_result=False, lane=None):
        """uint8 [B, H, W, 3] -> (string[B], side_string[B], x_shape, y_shape, z_shape) —
        bmshj2018.py:219-240.

        `device_result=True`: the two string entries are finalized encoder handles whose bytes stay in
        HBM (`gen_ops.fetch_strings` / `gen_ops.device_strings`), nothing is read back and the call only
        enqueues work; `lane` (compression_amd.pipeline.Lane) puts the transforms and the coder on their
        own streams / compute units."""
        lane = lane or inline_lane()
        if x.dim() == 3:
            x = x[None]
        with lane.on("transform"):
            y = self.analysis_transform.unit(functional.image_to_unit(x, self.compute_dtype))
            z = self.hyper_analysis_transform(torch.abs(y))
            x_shape, y_shape, z_shape = tuple(x.shape[1:-1]), tuple(y.shape[1:-1]), tuple(z.shape[1:-1])
            z_hat = self.side_entropy_model.quantize(z)
            indexes = self.hyper_synthesis_transform(z_hat)[:, :y_shape[0], :y_shape[1], :]
        with lane.on("coder"):
            side_string = self.side_entropy_model.compress(z, device_result=device_result)
            string = self.entropy_model.compress(y, indexes, device_result=device_result)
        if device_result:
            # produced on the transform stream, read by the coder stream: alive as long as the strings
            string._keep += [y, indexes]
            side_string._keep += [z]
        return string, side_string, x_shape, y_shape, z_shape

    @torch.no_grad()
    def decompress(self, string, side_string, x_shape, y_shape, z_shape, defer_sanity=False, lane=None):
        """bmshj2018.py:242-264: the y stream can only be decoded after z (strict two-phase order).
        The strings may be the handles `compress(device_result=True)` returned; `defer_sanity=True`
        returns (x_hat, [ok_z, ok_y]) with the device-resident EntropyDecodeFinalize flags instead of
        reading them back."""
        lane = lane or inline_lane()
        with lane.on("coder"):
            z_hat = self.side_entropy_model.decompress(side_string, z_shape, defer_sanity=defer_sanity)
        ok = []
        if defer_sanity:
            z_hat, okz = z_hat
            ok.append(okz)
        with lane.on("transform"):
            indexes = self.hyper_synthesis_transform(z_hat)[:, :y_shape[0], :y_shape[1], :]
        with lane.on("coder"):
            y_hat = self.entropy_model.decompress(string, indexes, defer_sanity=defer_sanity)
        if defer_sanity:
            y_hat, oky = y_hat
            ok.append(oky)
        with lane.on("transform"):
            x_hat = functional.unit_to_image(self.synthesis_transform.unit(y_hat)[:, :x_shape[0], :x_shape[1], :])
            # produced on one stream, read on the other: alive until the caller drops x_hat (a tensor freed behind a
            # kernel of ANOTHER stream goes back to its own stream's pool while that kernel may still read it)
            x_hat._tfc_keep = (z_hat, indexes, y_hat)
        return (x_hat, ok) if defer_sanity else x_hat

    @torch.no_grad()
    def compress_many(self, xs):
        """compress() of several batches with ONE coder launch per stage and latent (the entropy models' compress_many:
        the pipelined lane kernels, whose serial chain is a few waves however many batches share the launch — the
        convolutions of other batches keep the rest of the chip): [(string handle, side handle, x_shape, y_shape,
        z_shape)], strings in HBM.  Same strings as compress() batch by batch."""
        ys, zs, idxs, shapes = [], [], [], []
        for x in xs:
            x = x if x.dim() == 4 else x[None]
            y = self.analysis_transform.unit(functional.image_to_unit(x, self.compute_dtype))
            z = self.hyper_analysis_transform(torch.abs(y))
            shapes.append((tuple(x.shape[1:-1]), tuple(y.shape[1:-1]), tuple(z.shape[1:-1])))
            z_hat = self.side_entropy_model.quantize(z)
            idxs.append(self.hyper_synthesis_transform(z_hat)[:, :y.shape[1], :y.shape[2], :])
            ys.append(y)
            zs.append(z)
        side = self.side_entropy_model.compress_many(zs)
        main = self.entropy_model.compress_many(ys, idxs)
        return [(h, sh) + shp for h, sh, shp in zip(main, side, shapes)]

    @torch.no_grad()
    def decompress_many(self, packed):
        """decompress() for the results of compress_many: ([x_hat per batch], [ok_z, ok_y] flags on the device).  Two
        phases like decompress(): all side latents, their hyper-synthesis, then all main latents."""
        packed = list(packed)
        if any(tuple(p[2:]) != tuple(packed[0][2:]) for p in packed):
            raise ValueError("decompress_many: all batches must have the same x / y / z shapes")
        z_hats, okz = self.side_entropy_model.decompress_many([p[1] for p in packed], packed[0][4])
        idxs = [self.hyper_synthesis_transform(z_hat)[:, :p[3][0], :p[3][1], :] for z_hat, p in zip(z_hats, packed)]
        y_hats, oky = self.entropy_model.decompress_many([p[0] for p in packed], idxs)
        outs = []
        for p, z_hat, ix, y_hat in zip(packed, z_hats, idxs, y_hats):
            x_hat = functional.unit_to_image(self.synthesis_transform.unit(y_hat)[:, :p[2][0], :p[2][1], :])
            x_hat._tfc_keep = (z_hat, ix, y_hat)
            outs.append(x_hat)
        return outs, [okz, oky]


if __name__ == "__main__":      # python -m compression_amd.models.bmshj2018 compress in.png out.tfci
    import sys

    from .codec_io import main
    sys.exit(main(BMSHJ2018Model))
