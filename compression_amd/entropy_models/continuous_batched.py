"""`ContinuousBatchedEntropyModel` (python/entropy_models/continuous_batched.py:33-436),
the TFC-2.x successor of `EntropyBottleneck`: quantisation, rate estimate, and range
coding with per-channel tables.  compress()/decompress() run on the HIP coder; the
quantise prologue / dequantise epilogue are fused into the coder's load/store."""
from __future__ import annotations

import ctypes as C
import functools

import numpy as np
import torch

from .. import _lib
from ..distributions import helpers
from ..ops import bottleneck_ops, gen_ops, math_ops, round_ops
from . import continuous_base

__all__ = ["ContinuousBatchedEntropyModel", "EntropyBottleneck"]

_DTYPE_CODE = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


class ContinuousBatchedEntropyModel(continuous_base.ContinuousEntropyModelBase):
    def __init__(self, prior=None, coding_rank=None, compression=False, stateless=False,
                 expected_grads=False, tail_mass=2 ** -8, range_coder_precision=12,
                 bottleneck_dtype=None, prior_shape=None, cdf=None, cdf_offset=None,
                 cdf_shapes=None, offset_heuristic=True, quantization_offset=None,
                 decode_sanity_check=True, laplace_tail_mass=0):
        if (prior is None) == (prior_shape is None):
            raise ValueError("Either `prior` or `prior_shape` must be provided.")
        if (prior is None) + (cdf_shapes is None) + (cdf is None) != 2:
            raise ValueError("Must provide exactly one of `prior`, `cdf`, or `cdf_shapes`.")
        if not compression and not (cdf is None and cdf_offset is None and cdf_shapes is None):
            raise ValueError("CDFs can't be provided with `compression=False`")
        if prior is not None and len(prior.event_shape):
            raise ValueError("`prior` must be a (batch of) scalar distribution(s).")
        super().__init__(coding_rank=coding_rank, compression=compression, stateless=stateless,
                         expected_grads=expected_grads, tail_mass=tail_mass,
                         bottleneck_dtype=bottleneck_dtype, laplace_tail_mass=laplace_tail_mass)
        self._prior = prior
        self._offset_heuristic = bool(offset_heuristic)
        self._prior_shape = torch.Size(prior_shape if prior is None else prior.batch_shape)
        if self.coding_rank < len(self.prior_shape):
            raise ValueError("`coding_rank` can't be smaller than `prior_shape`.")
        self.decode_sanity_check = decode_sanity_check
        self.fused = True   # quantise/dequantise inside the coder kernels

        if cdf_shapes is not None:
            assert isinstance(quantization_offset, bool) and self.compression
            quantization_offset = torch.zeros(tuple(self.prior_shape)) if quantization_offset else None
        elif quantization_offset is not None:
            quantization_offset = torch.as_tensor(quantization_offset)
        elif self.offset_heuristic and self.compression:
            if self._prior is None:
                raise ValueError("To use the offset heuristic, a `prior` needs to be provided.")
            quantization_offset = helpers.quantization_offset(self.prior)
            if bool(torch.all(quantization_offset == 0.0)):
                quantization_offset = None
            else:
                quantization_offset = quantization_offset.expand(tuple(self.prior_shape)).clone()
        if quantization_offset is None:
            self._quantization_offset = None
        else:
            q = quantization_offset.detach().to(self.bottleneck_dtype)
            if self.compression and not self.stateless:
                self.register_buffer("_quantization_offset", q)
            else:
                self._quantization_offset = q
        if self.compression:
            if cdf is None and cdf_shapes is None:
                cdf, cdf_offset = self._build_tables(self.prior, range_coder_precision,
                                                     offset=quantization_offset)
            self._init_compression(cdf, cdf_offset, cdf_shapes)

    prior_shape = property(lambda self: self._prior_shape)
    offset_heuristic = property(lambda self: self._offset_heuristic)

    @property
    def quantization_offset(self):
        if self._quantization_offset is not None:
            return self._quantization_offset
        if self.offset_heuristic and not self.compression:
            if self._prior is None:
                raise RuntimeError("To use the offset heuristic, a `prior` needs to be provided.")
            return helpers.quantization_offset(self.prior).to(self.bottleneck_dtype)
        return None

    def forward(self, bottleneck, training=True):
        """(bottleneck_perturbed, bits) — continuous_batched.py:291-322."""
        bottleneck = torch.as_tensor(bottleneck).to(self.bottleneck_dtype)
        base = getattr(self.prior, "base", None)
        ltm = bottleneck_ops.fused_tail_mass(self.laplace_tail_mass)
        if (training and ltm is not None
                and bottleneck_ops.fused_factorized_supported(base, bottleneck, self.coding_rank)):
            # one fused HIP kernel each way: noise add + likelihood (with the Laplace-mixture tail if the model
            # has one) + bits (csrc/factorized_bits.hip); expected_grads selects the backward kernel's
            # finite-difference input gradient
            noise = torch.rand_like(bottleneck) - 0.5
            return bottleneck_ops.factorized_bits(bottleneck, base, self.coding_rank, noise,
                                                  expected_grads=self.expected_grads, laplace_tail_mass=ltm)
        log_prob_fn = functools.partial(self._log_prob, self.prior)
        if training:
            log_probs, perturbed = math_ops.perturb_and_apply(
                log_prob_fn, bottleneck, expected_grads=self.expected_grads)
        else:
            perturbed = self.quantize(bottleneck)
            log_probs = log_prob_fn(perturbed)
        axes = tuple(range(-self.coding_rank, 0))
        bits = log_probs.sum(dim=axes) / (-float(np.log(2.0))) if axes else log_probs / (-float(np.log(2.0)))
        return perturbed, bits

    def quantize(self, bottleneck):
        bottleneck = torch.as_tensor(bottleneck).to(self.bottleneck_dtype)
        offset = self.quantization_offset
        if offset is not None:
            if self.compression and bottleneck.is_cuda:
                offset = self._device_tables(bottleneck.device)[2]     # cached: no copy (and no sync) per call
            else:
                offset = offset.to(bottleneck.device)
        return round_ops.round_st(bottleneck, offset)

    # ------------------------------------------------------------------ coding
    def _device_tables(self, device):
        """(cdf_offset, quantization offset as float32 [C], quantization offset as stored) on the device,
        uploaded once per table version."""
        cache = getattr(self, "_dev_cache", None)
        key = (self.cdf.data_ptr(), getattr(self.cdf, "_version", 0), str(device))
        if cache is None or cache[0] != key:
            off = self.cdf_offset.to(device).contiguous()
            q = self._quantization_offset
            qf = None if q is None else q.to(device, torch.float32).reshape(-1).contiguous()
            qn = None if q is None else q.to(device)
            object.__setattr__(self, "_dev_cache", (key, off, qf, qn))
            cache = self._dev_cache
        return cache[1], cache[2], cache[3]

    def compress(self, bottleneck, device_result=False):
        """continuous_batched.py:347-383.  Returns a numpy object array of `bytes`
        shaped like `bottleneck` minus the `coding_rank` innermost dimensions.

        `device_result=True`: nothing is read back — the call is enqueued on the current HIP stream and
        returns the finalized encoder handle, whose strings stay in HBM (`gen_ops.device_strings`,
        `gen_ops.fetch_strings`; `decompress` takes the handle in place of the strings).  Range errors are
        then reported by `fetch_strings` / `gen_ops.entropy_encode_status`.  So is a stream that outgrows the output
        slab such a call sizes without reading anything back (2 bytes per symbol, a quarter more when the tables have
        escape rows — i.e. more than ~16-20 bits per symbol on average over a whole stream, which only data far off
        the model's tables produces): "a stream outgrew its output slab".  The plain call (`device_result=False`)
        repeats itself with the bound that cannot be exceeded instead; call it for such data."""
        self._check_compression()
        device = _lib.require_device()
        bottleneck = torch.as_tensor(bottleneck).to(device, self.bottleneck_dtype).contiguous()
        shape = tuple(bottleneck.shape)
        batch_shape = shape[:len(shape) - self.coding_rank] if self.coding_rank else shape
        handle = gen_ops.create_range_encoder(batch_shape, self.cdf, deferred_errors=device_result)
        if handle.streams == 0:
            raise ValueError(f"`handle` is empty: handle.shape={list(batch_shape)}")
        channels = int(self.prior_shape.numel())
        elems = bottleneck.numel() // handle.streams
        cdf_offset, qoff, _ = self._device_tables(device)
        if self.fused and bottleneck.dtype in _DTYPE_CODE:
            handle._keep += [bottleneck, cdf_offset, qoff]
            quantized = self._quantized_call(bottleneck, qoff, cdf_offset, channels, elems)
            handle.record(quantized)
            quantized(handle)
        else:
            offset = self.quantization_offset
            if offset is not None:
                bottleneck = bottleneck - offset.to(device)
            symbols = torch.round(bottleneck).to(torch.int32)
            iid = shape[:len(shape) - len(self.prior_shape)] if len(self.prior_shape) else shape
            symbols = symbols.reshape(tuple(iid) + (-1,)) - cdf_offset
            handle = gen_ops.entropy_encode_channel(handle, symbols.contiguous())
        if device_result:
            handle.coder_inputs = (bottleneck, None)     # what the coder read (values, table indexes): for checkers
            return gen_ops.entropy_encode_finalize_device(handle)
        return gen_ops.entropy_encode_finalize(handle)

    def decompress(self, strings, broadcast_shape, defer_sanity=False):
        """continuous_batched.py:385-422: output shape = strings.shape + broadcast_shape +
        prior_shape.  `strings`: bytes container, or a handle from `compress(device_result=True)`.

        `defer_sanity=True`: nothing is read back; returns (values, ok) with `ok` the device-resident
        uint8 result of EntropyDecodeFinalize (1 = the end-of-stream check passed) for the caller to test
        once the stream has run."""
        self._check_compression()
        device = _lib.require_device()
        broadcast_shape = tuple(int(s) for s in broadcast_shape)
        handle = gen_ops.create_range_decoder(strings, self.cdf)
        channels = int(self.prior_shape.numel())
        out_shape = tuple(handle.shape) + broadcast_shape + tuple(self.prior_shape)
        elems = int(np.prod(broadcast_shape, dtype=np.int64)) * channels
        cdf_offset, qoff, _ = self._device_tables(device)
        if self.fused and self.bottleneck_dtype in _DTYPE_CODE:
            out = torch.empty(out_shape, dtype=self.bottleneck_dtype, device=device)
            _lib.check(_lib.lib().tfc_decoder_decode_dequantized(
                handle.ptr, None, out.data_ptr(), _DTYPE_CODE[self.bottleneck_dtype],
                None if qoff is None else qoff.data_ptr(), cdf_offset.data_ptr(), channels, elems,
                _lib.stream_ptr()))
            handle._keep.append(out)
        else:
            handle, symbols = gen_ops.entropy_decode_channel(
                handle, broadcast_shape + (channels,), torch.int32)
            out = (symbols + cdf_offset).reshape(out_shape).to(self.bottleneck_dtype)
            offset = self.quantization_offset
            if offset is not None:
                out = out + offset.to(device)
        if defer_sanity:
            ok = gen_ops.entropy_decode_finalize_device(handle)
            ok._tfc_handle = handle          # the decoder (and what it borrows) lives as long as its verdict
            return out, ok
        sanity = gen_ops.entropy_decode_finalize(handle)
        if self.decode_sanity_check and not bool(sanity.all()):
            raise RuntimeError("Sanity check failed.")
        return out

    @staticmethod
    def _quantized_call(bottleneck, qoff, cdf_offset, channels, elems):
        """The fused quantise + encode call of one handle as a closure over its inputs (what a deferred handle keeps to be
        coded again: gen_ops._retry_outgrown)."""
        def call(handle):
            handle._keep += [bottleneck, cdf_offset, qoff]
            _lib.check(_lib.lib().tfc_encoder_encode_quantized(
                handle.ptr, bottleneck.data_ptr(), _DTYPE_CODE[bottleneck.dtype],
                None if qoff is None else qoff.data_ptr(), cdf_offset.data_ptr(), channels, elems,
                _lib.stream_ptr()))
        return call

    # ------------------------------------------------------------------ several batches per launch
    def _symbols(self, bottleneck, cdf_offset, qoff_native):
        """int32 symbols of continuous_batched.py:370-380 as tensor ops (the `*_many` path hands the coder
        plain symbols: the lane-per-stream kernels' hand-scheduled blocks take int32 in channel mode)."""
        if qoff_native is not None:
            bottleneck = bottleneck - qoff_native
        symbols = torch.round(bottleneck.float()).to(torch.int32)
        iid = bottleneck.shape[:bottleneck.dim() - len(self.prior_shape)] if len(self.prior_shape) else bottleneck.shape
        return (symbols.reshape(tuple(iid) + (-1,)) - cdf_offset).reshape(bottleneck.shape).contiguous()

    def compress_many(self, bottlenecks):
        """compress() of several independent batches (same shape) with ONE coder launch: the batches'
        code streams share the lane-per-stream kernels' grid (tfc_encoder_encode_many), whose running time
        is set by the symbols per stream, not by the number of streams, until every SIMD holds a wave — the
        way a server that has several batches in flight fills the chip.  Nothing is read back: returns one
        finalized encoder handle per batch (`gen_ops.fetch_strings`, `decompress_many`).  Same strings as
        compress() batch by batch.  (Like `compress(device_result=True)`: a stream of more than ~16-20 bits per
        symbol outgrows the speculative slab; `fetch_strings` then codes that handle again, synchronously.)"""
        self._check_compression()
        device = _lib.require_device()
        bottlenecks = [torch.as_tensor(b).to(device, self.bottleneck_dtype).contiguous() for b in bottlenecks]
        if not bottlenecks:
            return []
        shape = tuple(bottlenecks[0].shape)
        if any(tuple(b.shape) != shape for b in bottlenecks):
            raise ValueError("compress_many: all bottlenecks must have the same shape")
        batch_shape = shape[:len(shape) - self.coding_rank] if self.coding_rank else shape
        cdf_offset, qoff, qn = self._device_tables(device)
        n = len(bottlenecks)
        handles = gen_ops.create_range_encoders(n, batch_shape, self.cdf, mode="throughput", deferred_errors=True)
        if self.fused and bottlenecks[0].dtype in _DTYPE_CODE:
            # quantise prologue inside the library: one elementwise pass per batch, then ONE coding launch
            channels = int(self.prior_shape.numel())
            elems = bottlenecks[0].numel() // handles[0].streams
            hp = (C.c_void_p * n)(*[h.ptr for h in handles])
            yp = (C.c_void_p * n)(*[b.data_ptr() for b in bottlenecks])
            _lib.check(_lib.lib().tfc_encoder_encode_quantized_many(
                n, hp, yp, _DTYPE_CODE[bottlenecks[0].dtype], None if qoff is None else qoff.data_ptr(),
                cdf_offset.data_ptr(), channels, elems, _lib.stream_ptr()))
            keep = [[b, cdf_offset, qoff] for b in bottlenecks]
            for h, b in zip(handles, bottlenecks):
                h.record(self._quantized_call(b, qoff, cdf_offset, channels, elems))
        else:
            symbols = [self._symbols(b, cdf_offset, qn) for b in bottlenecks]
            handles = gen_ops.entropy_encode_channel_many(handles, symbols)
            keep = [[b, s] for b, s in zip(bottlenecks, symbols)]
        handles = gen_ops.entropy_encode_finalize_device_many(handles)
        for h, b, k in zip(handles, bottlenecks, keep):
            h.coder_inputs = (b, None)
            h._keep += k
        return handles

    def decompress_many(self, handles, broadcast_shape):
        """decompress() for the handles of compress_many (or any finalized encoder handles of one shape): one
        decoder launch for all of them; returns ([values per batch], ok) with `ok` the device-resident
        EntropyDecodeFinalize flags [len(handles), *batch_shape] — nothing is read back."""
        self._check_compression()
        device = _lib.require_device()
        handles = list(handles)
        if not handles:
            return [], None
        broadcast_shape = tuple(int(s) for s in broadcast_shape)
        channels = int(self.prior_shape.numel())
        cdf_offset, qoff, qn = self._device_tables(device)
        decoders = gen_ops.create_range_decoders(handles, self.cdf, mode="throughput")
        n = len(decoders)
        out_shape = tuple(handles[0].shape) + broadcast_shape + tuple(self.prior_shape)
        if self.fused and self.bottleneck_dtype in _DTYPE_CODE:
            outs = [torch.empty(out_shape, dtype=self.bottleneck_dtype, device=device) for _ in range(n)]
            elems = int(np.prod(broadcast_shape, dtype=np.int64)) * channels
            dp = (C.c_void_p * n)(*[d.ptr for d in decoders])
            yp = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
            _lib.check(_lib.lib().tfc_decoder_decode_dequantized_many(
                n, dp, yp, _DTYPE_CODE[self.bottleneck_dtype], None if qoff is None else qoff.data_ptr(),
                cdf_offset.data_ptr(), channels, elems, _lib.stream_ptr()))
            for d, o in zip(decoders, outs):
                d._keep += [o, cdf_offset, qoff]
        else:
            decoders, symbols = gen_ops.entropy_decode_channel_many(decoders, broadcast_shape + (channels,), torch.int32)
            outs = []
            for sym in symbols:
                out = (sym + cdf_offset).reshape(out_shape).to(self.bottleneck_dtype)
                if qn is not None:
                    out = out + qn
                outs.append(out)
        ok = gen_ops.entropy_decode_finalize_device_many(decoders)
        ok._tfc_handle = decoders
        return outs, ok

    def get_config(self):
        config = super().get_config()
        config.update(prior_shape=tuple(map(int, self.prior_shape)),
                      offset_heuristic=self.offset_heuristic,
                      quantization_offset=self.quantization_offset is not None)
        return config

    @classmethod
    def from_config(cls, config):
        config = dict(config)
        config["bottleneck_dtype"] = getattr(torch, config["bottleneck_dtype"])
        return cls(**config)


class EntropyBottleneck(ContinuousBatchedEntropyModel):
    """TFC-1.x name kept by the north star: a `ContinuousBatchedEntropyModel` over a
    `NoisyDeepFactorized(batch_shape=(filters,))` prior with coding_rank 3, which is what
    `models/bls2017.py:108,160` instantiates in the 2.x API."""

    def __init__(self, filters, coding_rank=3, **kwargs):
        from ..distributions import NoisyDeepFactorized
        super().__init__(prior=NoisyDeepFactorized(batch_shape=(int(filters),)),
                         coding_rank=coding_rank, **kwargs)
