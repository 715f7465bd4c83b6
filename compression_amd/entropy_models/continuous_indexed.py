"""Indexed (conditional) entropy models
(python/entropy_models/continuous_indexed.py:30-633)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib
from ..ops import bottleneck_ops, gen_ops, math_ops, round_ops
from . import continuous_base

__all__ = ["ContinuousIndexedEntropyModel", "LocationScaleIndexedEntropyModel"]

_DTYPE_CODE = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


class ContinuousIndexedEntropyModel(continuous_base.ContinuousEntropyModelBase):
    def __init__(self, prior_fn, index_ranges, parameter_fns, coding_rank, channel_axis=-1,
                 compression=False, stateless=False, expected_grads=False, tail_mass=2 ** -8,
                 range_coder_precision=12, bottleneck_dtype=None, prior_dtype=torch.float32,
                 decode_sanity_check=True, laplace_tail_mass=0):
        if not callable(prior_fn):
            raise TypeError("`prior_fn` must be a class or factory function.")
        for name, fn in parameter_fns.items():
            if not isinstance(name, str):
                raise TypeError("`parameter_fns` must have string keys.")
            if not callable(fn):
                raise TypeError(f"`parameter_fns['{name}']` must be callable.")
        super().__init__(coding_rank=coding_rank, compression=compression, stateless=stateless,
                         expected_grads=expected_grads, tail_mass=tail_mass,
                         bottleneck_dtype=bottleneck_dtype, laplace_tail_mass=laplace_tail_mass)
        self._index_ranges = tuple(int(r) for r in index_ranges)
        if not self.index_ranges:
            raise ValueError("`index_ranges` must have at least one element.")
        self._channel_axis = None if channel_axis is None else int(channel_axis)
        if self.channel_axis is None and len(self.index_ranges) > 1:
            raise ValueError("`channel_axis` can't be `None` for `len(index_ranges) > 1`.")
        self._prior_fn = prior_fn
        self._parameter_fns = dict(parameter_fns)
        self._prior_dtype = prior_dtype
        self.decode_sanity_check = decode_sanity_check
        self.fused = True
        if self.compression:
            if self.channel_axis is None:
                indexes = torch.arange(self.index_ranges[0], dtype=torch.int32)
            else:
                grids = torch.meshgrid(*[torch.arange(r, dtype=torch.int32) for r in self.index_ranges],
                                       indexing="ij")
                indexes = torch.stack(grids, dim=self.channel_axis)
            self._prior = self._make_prior(indexes)
            cdf, cdf_offset = self._build_tables(self.prior, range_coder_precision)
            self._init_compression(cdf, cdf_offset, None)

    index_ranges = property(lambda self: self._index_ranges)
    parameter_fns = property(lambda self: self._parameter_fns)
    prior_dtype = property(lambda self: self._prior_dtype)
    prior_fn = property(lambda self: self._prior_fn)
    channel_axis = property(lambda self: self._channel_axis)

    def _make_prior(self, indexes):
        indexes = indexes.to(self.prior_dtype)
        parameters = {k: f(indexes) for k, f in self.parameter_fns.items()}
        prior = self.prior_fn(**parameters)
        assert prior.dtype == self.prior_dtype
        if len(prior.event_shape):
            raise ValueError("`prior` must be a (batch of) scalar distribution(s).")
        return prior

    def _normalize_indexes(self, indexes):
        indexes = math_ops.lower_bound(indexes, 0)
        if self.channel_axis is None:
            bounds = self.index_ranges[0] - 1       # a number: a device tensor would cost a synchronising copy per call
        else:
            axes = [1] * indexes.dim()
            axes[self.channel_axis] = len(self.index_ranges)
            bounds = torch.tensor([s - 1 for s in self.index_ranges], dtype=indexes.dtype,
                                  device=indexes.device).reshape(axes)
        return math_ops.upper_bound(indexes, bounds)

    def _table_indexes(self, indexes):
        """The coder's int32 table indexes: `_normalize_indexes` + `_flatten_indexes` (continuous_indexed.py:272-296),
        as one kernel for a single index range over floating-point indexes."""
        if self.channel_axis is None:
            from ..layers import functional
            flat = functional.index_prepare(indexes, self.index_ranges[0])
            if flat is not None:
                return flat
        return self._flatten_indexes(self._normalize_indexes(indexes)).contiguous()

    def _flatten_indexes(self, indexes):
        indexes = indexes.to(torch.int32)
        if self.channel_axis is None:
            return indexes
        strides = np.cumprod((self.index_ranges + (1,))[::-1])[::-1][1:]
        strides = torch.tensor(strides.copy(), dtype=torch.int32, device=indexes.device)
        # elementwise: integer matmul / tensordot has no HIP kernel ("addmm_cuda" not implemented for 'Int')
        return (indexes.movedim(self.channel_axis, -1) * strides).sum(-1, dtype=torch.int32)

    def forward(self, bottleneck, indexes, training=True):
        bottleneck = torch.as_tensor(bottleneck).to(self.bottleneck_dtype)
        indexes = self._normalize_indexes(torch.as_tensor(indexes))
        ltm = bottleneck_ops.fused_tail_mass(self.laplace_tail_mass)
        fused = training and ltm is not None and bottleneck_ops.fused_noisy_normal_supported(
            self.prior_fn, self.parameter_fns, bottleneck, self.coding_rank)
        if fused:
            idx = indexes.to(self.prior_dtype)
            loc = self.parameter_fns["loc"](idx)
            # the Laplace component of the tail mixture sits at 0 of the UNSHIFTED bottleneck
            # (continuous_base.py:298-334): the kernel, which sees the shifted one, takes it only for loc == 0
            fused = ltm == 0 or not (torch.is_tensor(loc) or loc != 0)
        if fused:
            # NoisyNormal(loc, scale_fn(indexes)): one fused HIP kernel each way (csrc/noisy_normal_bits.hip);
            # the parameter functions stay differentiable tensor ops, their gradients flow through `scale`
            scale = torch.as_tensor(self.parameter_fns["scale"](idx), dtype=self.prior_dtype, device=bottleneck.device)
            shifted = bottleneck - loc if (torch.is_tensor(loc) or loc != 0) else bottleneck
            noise = torch.rand_like(bottleneck) - 0.5
            perturbed, bits = bottleneck_ops.noisy_normal_bits(shifted, scale, self.coding_rank, noise,
                                                               expected_grads=self.expected_grads,
                                                               laplace_tail_mass=ltm)
            if torch.is_tensor(loc) or loc != 0:
                perturbed = perturbed + loc
            return perturbed, bits
        if training:
            def log_prob_fn(perturbed, idx):
                return self._log_prob(self._make_prior(idx), perturbed)
            log_probs, perturbed = math_ops.perturb_and_apply(
                log_prob_fn, bottleneck, indexes, expected_grads=self.expected_grads)
        else:
            prior = self._make_prior(indexes)
            perturbed = self.quantize(bottleneck)
            log_probs = self._log_prob(prior, perturbed)
        axes = tuple(range(-self.coding_rank, 0))
        # coding_rank 0: nothing to reduce (torch's sum over an empty dim tuple reduces EVERYTHING)
        bits = (log_probs.sum(dim=axes) if axes else log_probs) / (-float(np.log(2.0)))
        return perturbed, bits

    def quantize(self, bottleneck):
        return round_ops.round_st(torch.as_tensor(bottleneck).to(self.bottleneck_dtype))

    def _device_offsets(self, device):
        """cdf_offset on the device, uploaded once per table version (a pageable H->D copy in every call
        would put a host synchronisation into the coding path)."""
        key = (self.cdf_offset.data_ptr(), getattr(self.cdf_offset, "_version", 0), str(device))
        cache = getattr(self, "_dev_offsets", None)
        if cache is None or cache[0] != key:
            object.__setattr__(self, "_dev_offsets", (key, self.cdf_offset.to(device).contiguous()))
            cache = self._dev_offsets
        return cache[1]

    def compress(self, bottleneck, indexes, device_result=False):
        """continuous_indexed.py:355-386.  `device_result=True`: see
        `ContinuousBatchedEntropyModel.compress` (nothing read back, returns the finalized handle)."""
        self._check_compression()
        device = _lib.require_device()
        bottleneck = torch.as_tensor(bottleneck).to(device, self.bottleneck_dtype).contiguous()
        flat = self._table_indexes(torch.as_tensor(indexes).to(device))
        shape = tuple(flat.shape)
        batch_shape = shape[:len(shape) - self.coding_rank] if self.coding_rank else shape
        cdf_offset = self._device_offsets(device)
        handle = gen_ops.create_range_encoder(batch_shape, self.cdf, deferred_errors=device_result)
        if handle.streams == 0:
            raise ValueError(f"`handle` is empty: handle.shape={list(batch_shape)}")
        if self.fused and bottleneck.dtype in _DTYPE_CODE:
            elems = flat.numel() // handle.streams
            handle._keep += [bottleneck, flat, cdf_offset]
            quantized = self._quantized_call(bottleneck, flat, cdf_offset, elems)
            handle.record(quantized)
            quantized(handle)
        else:
            symbols = torch.round(bottleneck).to(torch.int32) - cdf_offset[flat.long()]
            handle = gen_ops.entropy_encode_index(handle, flat, symbols.contiguous())
        if device_result:
            handle.coder_inputs = (bottleneck, flat)     # what the coder read (values, table indexes): for checkers
            return gen_ops.entropy_encode_finalize_device(handle)
        return gen_ops.entropy_encode_finalize(handle)

    def decompress(self, strings, indexes, defer_sanity=False):
        """continuous_indexed.py:388-417.  `strings` may be a handle from `compress(device_result=True)`;
        `defer_sanity=True` returns (values, ok) without reading anything back (see
        `ContinuousBatchedEntropyModel.decompress`)."""
        self._check_compression()
        device = _lib.require_device()
        flat = self._table_indexes(torch.as_tensor(indexes).to(device))
        shape = tuple(flat.shape)
        decode_shape = shape[len(shape) - self.coding_rank:] if self.coding_rank else ()
        cdf_offset = self._device_offsets(device)
        handle = gen_ops.create_range_decoder(strings, self.cdf)
        if tuple(handle.shape) + tuple(decode_shape) != shape:
            raise ValueError(
                "'index' shape should match 'handle' shape + 'shape': "
                f"index.shape={list(shape)}, handle.shape={list(handle.shape)}, shape={list(decode_shape)}")
        if self.fused and self.bottleneck_dtype in _DTYPE_CODE:
            out = torch.empty(shape, dtype=self.bottleneck_dtype, device=device)
            elems = flat.numel() // handle.streams
            handle._keep += [flat, out, cdf_offset]
            _lib.check(_lib.lib().tfc_decoder_decode_dequantized(
                handle.ptr, flat.data_ptr(), out.data_ptr(), _DTYPE_CODE[self.bottleneck_dtype],
                None, cdf_offset.data_ptr(), 0, elems, _lib.stream_ptr()))
        else:
            handle, symbols = gen_ops.entropy_decode_index(handle, flat, decode_shape, torch.int32)
            out = (symbols + cdf_offset[flat.long()]).to(self.bottleneck_dtype)
        if defer_sanity:
            ok = gen_ops.entropy_decode_finalize_device(handle)
            ok._tfc_handle = handle
            return out, ok
        sanity = gen_ops.entropy_decode_finalize(handle)
        if self.decode_sanity_check and not bool(sanity.all()):
            raise RuntimeError("Sanity check failed.")
        return out

    @staticmethod
    def _quantized_call(bottleneck, flat, cdf_offset, elems):
        """The fused quantise + indexed encode call of one handle as a closure over its inputs (what a deferred handle
        keeps to be coded again: gen_ops._retry_outgrown)."""
        def call(handle):
            handle._keep += [bottleneck, flat, cdf_offset]
            _lib.check(_lib.lib().tfc_encoder_encode_quantized_indexed(
                handle.ptr, bottleneck.data_ptr(), _DTYPE_CODE[bottleneck.dtype], flat.data_ptr(),
                cdf_offset.data_ptr(), elems, _lib.stream_ptr()))
        return call

    def compress_many(self, bottlenecks, indexes):
        """compress() of several independent batches (same shapes) with ONE coder launch per stage
        (tfc_encoder_encode_quantized_indexed_many: the pipelined lane kernels quantise and look the tables up in
        their parallel expansion pass; the serial chain of all batches' streams is one small grid).  Nothing is read
        back: one finalized encoder handle per batch (`gen_ops.fetch_strings`, `decompress_many`); same strings as
        compress() batch by batch.  `indexes`: one tensor per batch (continuous_indexed.py:355-386)."""
        self._check_compression()
        device = _lib.require_device()
        bottlenecks = [torch.as_tensor(b).to(device, self.bottleneck_dtype).contiguous() for b in bottlenecks]
        flats = [self._table_indexes(torch.as_tensor(i).to(device)) for i in indexes]
        if not bottlenecks:
            return []
        if len(flats) != len(bottlenecks):
            raise ValueError("compress_many: one index tensor per bottleneck")
        shape = tuple(flats[0].shape)
        if any(tuple(f.shape) != shape for f in flats) or any(tuple(b.shape) != tuple(bottlenecks[0].shape) for b in bottlenecks):
            raise ValueError("compress_many: all batches must have the same shape")
        batch_shape = shape[:len(shape) - self.coding_rank] if self.coding_rank else shape
        cdf_offset = self._device_offsets(device)
        n = len(bottlenecks)
        handles = gen_ops.create_range_encoders(n, batch_shape, self.cdf, mode="throughput", deferred_errors=True)
        if handles[0].streams == 0:
            raise ValueError(f"`handle` is empty: handle.shape={list(batch_shape)}")
        if self.fused and bottlenecks[0].dtype in _DTYPE_CODE:
            elems = flats[0].numel() // handles[0].streams
            hp = (C.c_void_p * n)(*[h.ptr for h in handles])
            yp = (C.c_void_p * n)(*[b.data_ptr() for b in bottlenecks])
            ip = (C.c_void_p * n)(*[f.data_ptr() for f in flats])
            _lib.check(_lib.lib().tfc_encoder_encode_quantized_indexed_many(
                n, hp, yp, _DTYPE_CODE[bottlenecks[0].dtype], ip, cdf_offset.data_ptr(), elems, _lib.stream_ptr()))
            for h, b, f in zip(handles, bottlenecks, flats):
                h.record(self._quantized_call(b, f, cdf_offset, elems))
        else:
            for k in range(n):
                symbols = torch.round(bottlenecks[k]).to(torch.int32) - cdf_offset[flats[k].long()]
                handles[k] = gen_ops.entropy_encode_index(handles[k], flats[k], symbols.contiguous())
        handles = gen_ops.entropy_encode_finalize_device_many(handles)
        for h, b, f in zip(handles, bottlenecks, flats):
            h.coder_inputs = (b, f)
            h._keep += [b, f, cdf_offset]
        return handles

    def decompress_many(self, handles, indexes):
        """decompress() for the handles of compress_many: one decoder launch per stage for all of them; returns
        ([values per batch], ok) with `ok` the device-resident EntropyDecodeFinalize flags — nothing is read back."""
        self._check_compression()
        device = _lib.require_device()
        handles = list(handles)
        if not handles:
            return [], None
        flats = [self._table_indexes(torch.as_tensor(i).to(device)) for i in indexes]
        if len(flats) != len(handles):
            raise ValueError(f"decompress_many: one index tensor per handle ({len(flats)} index tensors, {len(handles)} handles)")
        shape = tuple(flats[0].shape)
        if any(tuple(f.shape) != shape for f in flats):
            raise ValueError("decompress_many: all index tensors must have the same shape")
        decode_shape = shape[len(shape) - self.coding_rank:] if self.coding_rank else ()
        cdf_offset = self._device_offsets(device)
        decoders = gen_ops.create_range_decoders(handles, self.cdf, mode="throughput")
        n = len(decoders)
        for d in decoders:
            if d.streams == 0:
                raise ValueError(f"`handle` is empty: handle.shape={list(d.shape)}")
            if tuple(d.shape) + tuple(decode_shape) != shape:
                raise ValueError(
                    "'index' shape should match 'handle' shape + 'shape': "
                    f"index.shape={list(shape)}, handle.shape={list(d.shape)}, shape={list(decode_shape)}")
        if self.fused and self.bottleneck_dtype in _DTYPE_CODE:
            outs = [torch.empty(shape, dtype=self.bottleneck_dtype, device=device) for _ in range(n)]
            elems = flats[0].numel() // decoders[0].streams
            dp = (C.c_void_p * n)(*[d.ptr for d in decoders])
            ip = (C.c_void_p * n)(*[f.data_ptr() for f in flats])
            yp = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
            _lib.check(_lib.lib().tfc_decoder_decode_dequantized_indexed_many(
                n, dp, ip, yp, _DTYPE_CODE[self.bottleneck_dtype], cdf_offset.data_ptr(), elems, _lib.stream_ptr()))
            for d, f, o in zip(decoders, flats, outs):
                d._keep += [f, o, cdf_offset]
        else:
            outs = []
            for k in range(n):
                decoders[k], symbols = gen_ops.entropy_decode_index(decoders[k], flats[k], decode_shape, torch.int32)
                outs.append((symbols + cdf_offset[flats[k].long()]).to(self.bottleneck_dtype))
        ok = gen_ops.entropy_decode_finalize_device_many(decoders)
        ok._tfc_handle = decoders
        return outs, ok

    def get_config(self):
        raise NotImplementedError("Serializing indexed entropy models is not yet implemented.")


class LocationScaleIndexedEntropyModel(ContinuousIndexedEntropyModel):
    """continuous_indexed.py:431-633: `num_scales` scale tables, location handled by
    shifting the bottleneck."""

    def __init__(self, prior_fn, num_scales, scale_fn, coding_rank, compression=False,
                 stateless=False, expected_grads=False, tail_mass=2 ** -8, range_coder_precision=12,
                 bottleneck_dtype=None, prior_dtype=torch.float32, laplace_tail_mass=0):
        num_scales = int(num_scales)
        super().__init__(
            prior_fn=prior_fn, index_ranges=(num_scales,),
            parameter_fns=dict(loc=lambda _: 0.0, scale=scale_fn),
            coding_rank=coding_rank, channel_axis=None, compression=compression,
            stateless=stateless, expected_grads=expected_grads, tail_mass=tail_mass,
            range_coder_precision=range_coder_precision, bottleneck_dtype=bottleneck_dtype,
            prior_dtype=prior_dtype, laplace_tail_mass=laplace_tail_mass)

    def forward(self, bottleneck, scale_indexes, loc=None, training=True):
        if loc is None:
            return super().forward(bottleneck, scale_indexes, training=training)
        perturbed, bits = super().forward(bottleneck - loc, scale_indexes, training=training)
        return perturbed + loc, bits

    def quantize(self, bottleneck, loc=None):
        return round_ops.round_st(torch.as_tensor(bottleneck).to(self.bottleneck_dtype), loc)

    def compress(self, bottleneck, scale_indexes, loc=None, device_result=False):
        if loc is not None:
            bottleneck = bottleneck - loc
        return super().compress(bottleneck, scale_indexes, device_result=device_result)

    def decompress(self, strings, scale_indexes, loc=None, defer_sanity=False):
        values = super().decompress(strings, scale_indexes, defer_sanity=defer_sanity)
        ok = None
        if defer_sanity:
            values, ok = values
        if loc is not None:
            values = values + loc
        return (values, ok) if defer_sanity else values
