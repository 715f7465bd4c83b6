"""Universal-quantisation entropy models (python/entropy_models/universal.py:30-603;
"Universally Quantized Neural Compression", Agustsson & Theis): quantisation is subtractive dither —
round(y - o) + o with a pseudo-random offset o the decoder can regenerate — and the range-coding tables are
built once per offset level.  compress() / decompress() are the indexed HIP coder ops with the offset level
as one more index dimension.

The shared randomness.  The reference draws the offset levels with
`tf.random.stateless_uniform(shape, seed=(1234, 1234), minval=0, maxval=num_noise_levels, dtype=int32)`
(universal.py:30-41).  `stateless_offset_indexes` below restates that op — Philox-4x32-10; key and counter
scrambled from the seed as TensorFlow's stateless ops do (one Philox block under the fixed key
0x3ec8f720, 0x02461e29 with the seed words as counter); one 32-bit draw per element in flat order, reduced
modulo the range — from TensorFlow's published algorithm.  TensorFlow is not available here, so this stream
is UNPINNED against the reference: encoder and decoder of this package agree with each other (that is what
the tests check); interoperability of the strings with the reference's additionally needs the stream to be
confirmed on a machine that has TensorFlow."""
from __future__ import annotations

import functools

import numpy as np
import torch

from .. import _lib
from ..ops import gen_ops, math_ops
from . import continuous_base

__all__ = ["UniversalBatchedEntropyModel", "UniversalIndexedEntropyModel", "stateless_offset_indexes"]

_M32 = np.uint64(0xFFFFFFFF)


def _philox4x32(counter, key, rounds=10):
    """Philox-4x32 on arrays of counters: counter [n, 4] uint32, key (k0, k1)."""
    c = counter.astype(np.uint64)
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    for _ in range(rounds):
        p0 = np.uint64(0xD2511F53) * c[:, 0]
        p1 = np.uint64(0xCD9E8D57) * c[:, 2]
        c = np.stack([((p1 >> np.uint64(32)) ^ c[:, 1] ^ k0) & _M32, p1 & _M32,
                      ((p0 >> np.uint64(32)) ^ c[:, 3] ^ k1) & _M32, p0 & _M32], axis=1)
        k0 = (k0 + np.uint64(0x9E3779B9)) & _M32
        k1 = (k1 + np.uint64(0xBB67AE85)) & _M32
    return c.astype(np.uint32)


# TensorFlow's seed scrambling for stateless ops (core/kernels/stateless_random_ops.cc, GenerateKey): ONE
# Philox block under a fixed key with the two seed words as the counter; its first two output words become
# the key, its last two the upper counter words, and the lower counter words count 4-draw blocks.
_SCRAMBLE_KEY = (0x3EC8F720, 0x02461E29)


@functools.lru_cache(maxsize=32)
def _stateless_uniform_int(n, seed, maxval):
    """`n` draws of stateless_uniform(seed=seed, minval=0, maxval=maxval, dtype=int32) in flat order."""
    s0, s1 = int(seed[0]) & 0xFFFFFFFFFFFFFFFF, int(seed[1]) & 0xFFFFFFFFFFFFFFFF
    seed_ctr = np.array([[s0 & 0xFFFFFFFF, s0 >> 32, s1 & 0xFFFFFFFF, s1 >> 32]], np.uint32)
    mix = _philox4x32(seed_ctr, _SCRAMBLE_KEY)[0]
    key = (int(mix[0]), int(mix[1]))
    blocks = (n + 3) // 4
    idx = np.arange(blocks, dtype=np.uint64)
    ctr = np.zeros((blocks, 4), np.uint32)
    ctr[:, 0] = (idx & _M32).astype(np.uint32)          # Skip(block): 128-bit add to a counter whose low
    ctr[:, 1] = (idx >> np.uint64(32)).astype(np.uint32)  # 64 bits start at zero
    ctr[:, 2] = mix[2]
    ctr[:, 3] = mix[3]
    draws = _philox4x32(ctr, key).reshape(-1)[:n]
    # UniformDistribution<PhiloxRandom, int32>: lo + draw % range, one 32-bit draw per element
    return (draws % np.uint32(maxval)).astype(np.int32)


def stateless_offset_indexes(shape, num_noise_levels):
    """int32 offset levels in [0, num_noise_levels) for a tensor of `shape` (universal.py:30-41)."""
    n = int(np.prod(shape)) if len(shape) else 1
    return torch.from_numpy(_stateless_uniform_int(n, (1234, 1234), int(num_noise_levels)).copy()).reshape(tuple(shape))


_NP_DTYPE = {torch.float32: np.float32, torch.float64: np.float64, torch.float16: np.float16}


def _offset_indexes_to_offset(offset_indexes, num_noise_levels, dtype):
    """(k + 1) / (L + 1) - 1/2 (universal.py:44-46), in the arithmetic type the reference's expression
    has: float64 for integer offset indexes (TensorFlow's `/` on int32 is a float64 true division), the
    tensor's own type for floating-point ones (the indexed model casts the drawn levels to the dtype of the
    caller's indexes, universal.py:40), then cast to `dtype`.

    There are only L distinct values: they are computed ONCE on the host with IEEE true division and
    gathered by level, so that encoder and decoder get bit-identical dithers whatever device each runs on
    (a device kernel may turn `x / c` into `x * (1 / c)`, one ulp off — enough to flip a rounding)."""
    arith = offset_indexes.dtype if offset_indexes.is_floating_point() else torch.float64
    if arith in _NP_DTYPE:
        k = np.arange(num_noise_levels, dtype=_NP_DTYPE[arith])
        one, half = k.dtype.type(1), k.dtype.type(0.5)
        table = torch.from_numpy((k + one) / k.dtype.type(num_noise_levels + 1) - half)
    else:                                   # bfloat16 levels: torch's CPU arithmetic (true division as well)
        k = torch.arange(num_noise_levels, dtype=arith)
        table = (k + 1) / (num_noise_levels + 1) - 0.5
    table = table.to(dtype).to(offset_indexes.device)
    return table[offset_indexes.long().clamp(0, num_noise_levels - 1)]


def _range_coding_offsets(num_noise_levels, prior_rank, dtype):
    """Offsets the tables are built for, shaped [L, 1, ..., 1] (universal.py:54-61: `tf.range(L, dtype=dtype)`,
    i.e. the levels AND the arithmetic are in `dtype`, the bottleneck's)."""
    k = torch.arange(num_noise_levels, dtype=dtype).reshape((-1,) + (1,) * prior_rank)
    return _offset_indexes_to_offset(k, num_noise_levels, dtype)


class UniversalBatchedEntropyModel(continuous_base.ContinuousEntropyModelBase):
    """universal.py:64-305."""

    def __init__(self, prior, coding_rank, compression=False, laplace_tail_mass=0.0, expected_grads=False,
                 tail_mass=2 ** -8, range_coder_precision=12, bottleneck_dtype=None, num_noise_levels=15,
                 stateless=False, decode_sanity_check=True):
        if len(prior.event_shape):
            raise ValueError("`prior` must be a (batch of) scalar distribution(s).")
        super().__init__(coding_rank=coding_rank, compression=compression, stateless=stateless,
                         expected_grads=expected_grads, tail_mass=tail_mass,
                         bottleneck_dtype=bottleneck_dtype, laplace_tail_mass=laplace_tail_mass)
        self._prior = prior
        self._num_noise_levels = int(num_noise_levels)
        if self.coding_rank < len(self.prior_shape):
            raise ValueError("`coding_rank` can't be smaller than `prior_shape`.")
        self.decode_sanity_check = decode_sanity_check
        if self.compression:
            offset = _range_coding_offsets(self._num_noise_levels, len(self.prior_shape), self.bottleneck_dtype)
            cdf, cdf_offset = self._build_tables(self.prior, range_coder_precision, offset=offset)
            self._init_compression(cdf, cdf_offset, None)

    num_noise_levels = property(lambda self: self._num_noise_levels)

    @property
    def prior_shape(self):
        return torch.Size(self.prior.batch_shape)

    def _compute_indexes_and_offset(self, broadcast_shape):
        """Table index (offset level, prior element) and dither of every element of one coding unit's
        [broadcast_shape + prior_shape] block (universal.py:163-187): one offset level per prior-sized
        group, shared by encoder and decoder."""
        prior_size = int(self.prior_shape.numel())
        broadcast_shape = tuple(int(s) for s in broadcast_shape)
        offset_indexes = stateless_offset_indexes(broadcast_shape + (prior_size,), self._num_noise_levels)
        offset = _offset_indexes_to_offset(offset_indexes, self._num_noise_levels, self.bottleneck_dtype)
        indexes = offset_indexes * prior_size + torch.arange(prior_size, dtype=torch.int32)
        full = broadcast_shape + tuple(self.prior_shape)
        return indexes.reshape(full).to(torch.int32), offset.reshape(full)

    def _split(self, shape):
        shape = tuple(shape)
        batch, coding = shape[:len(shape) - self.coding_rank], shape[len(shape) - self.coding_rank:]
        return batch, coding, coding[:self.coding_rank - len(self.prior_shape)]

    def forward(self, bottleneck, training=True):
        """(bottleneck_perturbed, bits) — universal.py:189-227."""
        bottleneck = torch.as_tensor(bottleneck).to(self.bottleneck_dtype)
        log_prob_fn = functools.partial(self._log_prob, self.prior)
        if training:
            log_probs, perturbed = math_ops.perturb_and_apply(log_prob_fn, bottleneck, expected_grads=self.expected_grads)
        else:
            # H(round(bottleneck - noise) | noise)
            _, _, broadcast_shape = self._split(bottleneck.shape)
            _, offset = self._compute_indexes_and_offset(broadcast_shape)
            offset = offset.to(bottleneck.device)
            perturbed = torch.round(bottleneck - offset) + offset
            log_probs = log_prob_fn(perturbed)
        axes = tuple(range(-self.coding_rank, 0))
        bits = (log_probs.sum(dim=axes) if axes else log_probs) / (-float(np.log(2.0)))
        return perturbed, bits

    def _coder_inputs(self, bottleneck):
        """(symbols, table indexes) handed to EntropyEncodeIndex for `bottleneck` — universal.py:251-262:
        symbols = int32(round(bottleneck - offset)) - cdf_offset[indexes]; both int32, shaped like
        `bottleneck`, on its device."""
        device = bottleneck.device
        _, _, broadcast_shape = self._split(bottleneck.shape)
        indexes, offset = self._compute_indexes_and_offset(broadcast_shape)
        indexes, offset = indexes.to(device), offset.to(device)
        symbols = torch.round(bottleneck - offset).to(torch.int32) - self.cdf_offset.to(device)[indexes.long()]
        return symbols.contiguous(), torch.broadcast_to(indexes, symbols.shape).contiguous()

    def compress(self, bottleneck):
        """universal.py:229-266."""
        self._check_compression()
        device = _lib.require_device()
        bottleneck = torch.as_tensor(bottleneck).to(device, self.bottleneck_dtype)
        batch_shape, _, _ = self._split(bottleneck.shape)
        symbols, indexes = self._coder_inputs(bottleneck)
        handle = gen_ops.create_range_encoder(batch_shape, self.cdf)
        if handle.streams == 0:
            raise ValueError(f"`handle` is empty: handle.shape={list(batch_shape)}")
        handle = gen_ops.entropy_encode_index(handle, indexes, symbols)
        return gen_ops.entropy_encode_finalize(handle)

    def decompress(self, strings, broadcast_shape):
        """universal.py:268-300: `strings.shape + broadcast_shape + prior_shape`."""
        self._check_compression()
        device = _lib.require_device()
        strings = np.asarray(strings, dtype=object)
        broadcast_shape = tuple(int(s) for s in broadcast_shape)
        decode_shape = broadcast_shape + tuple(self.prior_shape)
        output_shape = tuple(strings.shape) + decode_shape
        indexes, offset = self._compute_indexes_and_offset(broadcast_shape)
        indexes, offset = indexes.to(device), offset.to(device)
        handle = gen_ops.create_range_decoder(strings, self.cdf)
        handle, symbols = gen_ops.entropy_decode_index(
            handle, torch.broadcast_to(indexes, output_shape).contiguous(), decode_shape, torch.int32)
        sanity = gen_ops.entropy_decode_finalize(handle)
        if self.decode_sanity_check and not bool(sanity.all()):
            raise RuntimeError("Sanity check failed.")
        symbols = symbols + self.cdf_offset.to(device)[indexes.long()]
        return symbols.to(self.bottleneck_dtype) + offset

    def get_config(self):
        raise NotImplementedError()


class UniversalIndexedEntropyModel(continuous_base.ContinuousEntropyModelBase):
    """universal.py:307-603.  `index_ranges[0]` is the number of offset levels; `indexes` carry the other
    dimensions in their last axis."""

    def __init__(self, prior_fn, index_ranges, parameter_fns, coding_rank, compression=False, dtype=torch.float32,
                 laplace_tail_mass=0.0, expected_grads=False, tail_mass=2 ** -8, range_coder_precision=12,
                 bottleneck_dtype=None, stateless=False, num_noise_levels=15, decode_sanity_check=True,
                 prior_dtype=None):
        if coding_rank <= 0:
            raise ValueError("`coding_rank` must be larger than 0.")
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
        self._num_noise_levels = int(num_noise_levels)
        # the offset level is the FIRST index dimension (universal.py:397-404)
        self._index_ranges = (self._num_noise_levels,) + tuple(int(r) for r in index_ranges)
        if len(self._index_ranges) < 2:
            raise ValueError("`index_ranges` must have at least one element.")
        self._prior_fn = prior_fn
        self._parameter_fns = dict(parameter_fns)
        self._prior_dtype = prior_dtype or dtype
        self.decode_sanity_check = decode_sanity_check
        if self.compression:
            grids = torch.meshgrid(*[torch.arange(r, dtype=torch.int32) for r in self.index_ranges_without_offsets],
                                   indexing="ij")
            indexes = torch.stack(grids, dim=-1)
            self._prior = self._make_prior(indexes)
            offset = _range_coding_offsets(self._num_noise_levels, len(self.prior.batch_shape), self.bottleneck_dtype)
            cdf, cdf_offset = self._build_tables(self.prior, range_coder_precision, offset=offset)
            self._init_compression(cdf, cdf_offset, None)

    index_ranges = property(lambda self: self._index_ranges)
    index_ranges_without_offsets = property(lambda self: self._index_ranges[1:])
    parameter_fns = property(lambda self: self._parameter_fns)
    prior_dtype = property(lambda self: self._prior_dtype)
    prior_fn = property(lambda self: self._prior_fn)
    num_noise_levels = property(lambda self: self._num_noise_levels)

    def _make_prior(self, indexes):
        indexes = indexes.to(self.prior_dtype)
        return self.prior_fn(**{k: f(indexes) for k, f in self.parameter_fns.items()})

    def _flatten_indexes(self, indexes):
        indexes = indexes.to(torch.int32)
        strides = np.cumprod((self.index_ranges + (1,))[::-1])[::-1][1:]
        strides = torch.tensor(strides.copy(), dtype=torch.int32, device=indexes.device)
        return (indexes * strides).sum(-1, dtype=torch.int32)

    def _normalize_indexes(self, indexes):
        n = indexes.shape[-1]
        ranges = self.index_ranges if n == len(self.index_ranges) else self.index_ranges_without_offsets
        if n != len(ranges):
            raise ValueError(f"the last dimension of `indexes` must be {len(self.index_ranges_without_offsets)}"
                             f" (or {len(self.index_ranges)} with offsets), got {n}")
        indexes = math_ops.lower_bound(indexes, 0)
        bounds = torch.tensor([r - 1 for r in ranges], dtype=indexes.dtype, device=indexes.device)
        return math_ops.upper_bound(indexes, bounds.reshape((1,) * (indexes.dim() - 1) + (n,)))

    def _add_offset_indexes(self, indexes):
        """universal.py:30-41: one offset level per element, prepended as index dimension 0."""
        off = stateless_offset_indexes(tuple(indexes.shape[:-1]), self._num_noise_levels).to(indexes.device)
        return torch.cat((off.to(indexes.dtype)[..., None], indexes), dim=-1)

    def _offset_from_indexes(self, indexes_with_offsets):
        return _offset_indexes_to_offset(indexes_with_offsets[..., 0], self._num_noise_levels, self.bottleneck_dtype)

    def forward(self, bottleneck, indexes, training=True):
        """universal.py:482-532."""
        bottleneck = torch.as_tensor(bottleneck).to(self.bottleneck_dtype)
        indexes = self._normalize_indexes(torch.as_tensor(indexes))
        if training:
            def log_prob_fn(perturbed, idx):
                return self._log_prob(self._make_prior(idx), perturbed)
            log_probs, perturbed = math_ops.perturb_and_apply(log_prob_fn, bottleneck, indexes,
                                                              expected_grads=self.expected_grads)
        else:
            prior = self._make_prior(indexes)
            offset = self._offset_from_indexes(self._add_offset_indexes(indexes)).to(bottleneck.device)
            perturbed = torch.round(bottleneck - offset) + offset
            log_probs = self._log_prob(prior, perturbed)
        axes = tuple(range(-self.coding_rank, 0))
        bits = log_probs.sum(dim=axes) / (-float(np.log(2.0)))
        return perturbed, bits

    def _coder_inputs(self, bottleneck, indexes):
        """(symbols, flat table indexes) handed to EntropyEncodeIndex — universal.py:557-565: offset level
        prepended to `indexes`, normalised, flattened; symbols = int32(round(bottleneck - offset)) -
        cdf_offset[flat]."""
        device = bottleneck.device
        indexes = self._normalize_indexes(self._add_offset_indexes(torch.as_tensor(indexes).to(device)))
        flat = self._flatten_indexes(indexes).contiguous()
        offset = self._offset_from_indexes(indexes)
        symbols = torch.round(bottleneck - offset).to(torch.int32) - self.cdf_offset.to(device)[flat.long()]
        return symbols.contiguous(), flat

    def compress(self, bottleneck, indexes):
        """universal.py:534-568."""
        self._check_compression()
        device = _lib.require_device()
        bottleneck = torch.as_tensor(bottleneck).to(device, self.bottleneck_dtype)
        symbols, flat = self._coder_inputs(bottleneck, indexes)
        batch_shape = tuple(flat.shape[:flat.dim() - self.coding_rank])
        handle = gen_ops.create_range_encoder(batch_shape, self.cdf)
        if handle.streams == 0:
            raise ValueError(f"`handle` is empty: handle.shape={list(batch_shape)}")
        handle = gen_ops.entropy_encode_index(handle, flat, symbols)
        return gen_ops.entropy_encode_finalize(handle)

    def decompress(self, strings, indexes):
        """universal.py:570-599."""
        self._check_compression()
        device = _lib.require_device()
        strings = np.asarray(strings, dtype=object)
        indexes = self._normalize_indexes(self._add_offset_indexes(torch.as_tensor(indexes).to(device)))
        flat = self._flatten_indexes(indexes).contiguous()
        decode_shape = tuple(flat.shape[flat.dim() - self.coding_rank:])
        handle = gen_ops.create_range_decoder(strings, self.cdf)
        handle, symbols = gen_ops.entropy_decode_index(handle, flat, decode_shape, torch.int32)
        sanity = gen_ops.entropy_decode_finalize(handle)
        if self.decode_sanity_check and not bool(sanity.all()):
            raise RuntimeError("Sanity check failed.")
        symbols = symbols + self.cdf_offset.to(device)[flat.long()]
        return symbols.to(self.bottleneck_dtype) + self._offset_from_indexes(indexes)

    def get_config(self):
        raise NotImplementedError()
