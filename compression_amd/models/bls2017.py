"""Ballé, Laparra, Simoncelli 2017 model (models/bls2017.py:55-190): analysis /
synthesis transforms + factorized-prior entropy bottleneck.  Class and layer names,
kernel supports and strides follow the reference; compress()/decompress() take a
BATCH of images (the reference's single-image signatures are the B = 1 case)."""
from __future__ import annotations

import torch

from .. import distributions, entropy_models, layers
from ..layers import functional
from ..pipeline import inline_lane

__all__ = ["AnalysisTransform", "SynthesisTransform", "BLS2017Model"]


class AnalysisTransform(torch.nn.Module):
    def __init__(self, num_filters):
        super().__init__()
        C = num_filters
        self.layer_0 = layers.SignalConv2D(C, (9, 9), corr=True, strides_down=4, padding="same_zeros",
                                           use_bias=True, activation=layers.GDN(), in_channels=3)
        self.layer_1 = layers.SignalConv2D(C, (5, 5), corr=True, strides_down=2, padding="same_zeros",
                                           use_bias=True, activation=layers.GDN(), in_channels=C)
        self.layer_2 = layers.SignalConv2D(C, (5, 5), corr=True, strides_down=2, padding="same_zeros",
                                           use_bias=False, activation=None, in_channels=C)

    def forward(self, x):
        return self.unit(x / 255.0)

    def unit(self, u):
        """The layers, on an image already scaled to [0, 1] (`functional.image_to_unit`)."""
        return self.layer_2(self.layer_1(self.layer_0(u)))


class SynthesisTransform(torch.nn.Module):
    def __init__(self, num_filters):
        super().__init__()
        C = num_filters
        self.layer_0 = layers.SignalConv2D(C, (5, 5), corr=False, strides_up=2, padding="same_zeros",
                                           use_bias=True, activation=layers.GDN(inverse=True), in_channels=C)
        self.layer_1 = layers.SignalConv2D(C, (5, 5), corr=False, strides_up=2, padding="same_zeros",
                                           use_bias=True, activation=layers.GDN(inverse=True), in_channels=C)
        self.layer_2 = layers.SignalConv2D(3, (9, 9), corr=False, strides_up=4, padding="same_zeros",
                                           use_bias=True, activation=None, in_channels=C)

    def forward(self, y):
        return self.unit(y) * 255.0

    def unit(self, y):
        """The layers, without the scaling to [0, 255] (`functional.unit_to_image` takes it with the rounding)."""
        return self.layer_2(self.layer_1(self.layer_0(y)))


class BLS2017Model(torch.nn.Module):
    # layout of the .tfci container: [string, x_shape, y_shape] (bls2017.py:164-176, 280-283)
    num_strings, num_packed = 1, 3

    def __init__(self, lmbda=0.01, num_filters=128, compute_dtype=torch.float32):
        super().__init__()
        self.lmbda = lmbda
        self.compute_dtype = compute_dtype
        self.analysis_transform = AnalysisTransform(num_filters)
        self.synthesis_transform = SynthesisTransform(num_filters)
        self.prior = distributions.NoisyDeepFactorized(batch_shape=(num_filters,))
        self.entropy_model = None

    def forward(self, x, training=True):
        """(loss, bpp, mse) — bls2017.py:109-125."""
        em = entropy_models.ContinuousBatchedEntropyModel(self.prior, coding_rank=3, compression=False,
                                                          bottleneck_dtype=self.compute_dtype)
        x = x.to(self.compute_dtype)
        y = self.analysis_transform(x)
        y_hat, bits = em(y, training=training)
        x_hat = self.synthesis_transform(y_hat.to(self.compute_dtype))
        num_pixels = x.shape[0] * x.shape[1] * x.shape[2]
        bpp = bits.sum() / num_pixels
        mse = torch.mean((x.float() - x_hat.float()) ** 2).to(bpp.dtype)
        return bpp + self.lmbda * mse, bpp, mse

    def init_compression(self):
        """What `fit()` does after training (bls2017.py:157-162): fix the range-coding tables."""
        self.entropy_model = entropy_models.ContinuousBatchedEntropyModel(
            self.prior, coding_rank=3, compression=True, bottleneck_dtype=self.compute_dtype)
        return self

    @torch.no_grad()
    def compress(self, x, device_result=False, lane=None):
        """x: uint8 [B, H, W, 3] (or [H, W, 3]) -> (strings[B], x_shape, y_shape) — bls2017.py:164-176.
        `device_result` / `lane`: see BMSHJ2018Model.compress (strings stay in HBM, nothing read back;
        transforms and coder on the lane's two streams)."""
        lane = lane or inline_lane()
        if x.dim() == 3:
            x = x[None]
        with lane.on("transform"):
            y = self.analysis_transform.unit(functional.image_to_unit(x, self.compute_dtype))
        with lane.on("coder"):
            string = self.entropy_model.compress(y, device_result=device_result)
        if device_result:
            string._keep.append(y)          # produced on the transform stream, read by the coder stream
        return string, tuple(x.shape[1:-1]), tuple(y.shape[1:-1])

    @torch.no_grad()
    def decompress(self, string, x_shape, y_shape, defer_sanity=False, lane=None):
        """-> uint8 [B, H, W, 3] — bls2017.py:178-190.  `defer_sanity=True`: (x_hat, [ok]) with the
        device-resident EntropyDecodeFinalize flags, nothing read back."""
        lane = lane or inline_lane()
        with lane.on("coder"):
            y_hat = self.entropy_model.decompress(string, y_shape, defer_sanity=defer_sanity)
        ok = []
        if defer_sanity:
            y_hat, oky = y_hat
            ok.append(oky)
        with lane.on("transform"):
            x_hat = functional.unit_to_image(self.synthesis_transform.unit(y_hat)[:, :x_shape[0], :x_shape[1], :])
            x_hat._tfc_keep = (y_hat,)      # produced on the coder stream, read here on the transform stream
        return (x_hat, ok) if defer_sanity else x_hat

    @torch.no_grad()
    def compress_many(self, xs):
        """compress() of several batches with one coder launch (ContinuousBatchedEntropyModel.compress_many):
        [(handle, x_shape, y_shape)] — the handles keep the strings in HBM."""
        ys = [self.analysis_transform.unit(functional.image_to_unit(x if x.dim() == 4 else x[None], self.compute_dtype))
              for x in xs]
        handles = self.entropy_model.compress_many(ys)
        return [(h, tuple(x.shape[-3:-1]), tuple(y.shape[1:-1])) for h, x, y in zip(handles, xs, ys)]

    @torch.no_grad()
    def decompress_many(self, packed):
        """decompress() for the results of compress_many: ([x_hat per batch], ok flags on the device)."""
        packed = list(packed)
        handles = [p[0] for p in packed]
        y_hats, ok = self.entropy_model.decompress_many(handles, packed[0][2])
        outs = []
        for (h, x_shape, y_shape), y_hat in zip(packed, y_hats):
            x_hat = functional.unit_to_image(self.synthesis_transform.unit(y_hat)[:, :x_shape[0], :x_shape[1], :])
            x_hat._tfc_keep = (y_hat,)
            outs.append(x_hat)
        return outs, ok


if __name__ == "__main__":      # python -m compression_amd.models.bls2017 compress in.png out.tfci
    import sys

    from .codec_io import main
    sys.exit(main(BLS2017Model))
