"""Op-level API: the callables the reference injects from its C++ plugin
(`python/ops/gen_ops.py:20-41` loading `cc/libtensorflow_compression.so`),
re-implemented on the C ABI of libtfc_hip.so (include/tfc_hip.h).

Same names, argument order and error substrings as the reference ops
(`cc/ops/range_coder_ops.cc`, `cc/ops/range_coding_ops.cc`,
`cc/ops/pmf_to_cdf_ops.cc`).  Tensors are torch tensors living in HBM;
"string tensors" are returned as numpy object arrays of `bytes` shaped like the
handle (plus a zero-copy device view on the handle).
"""
from __future__ import annotations

import ctypes as C
import threading
from collections import OrderedDict

import numpy as np
import torch

from .. import _lib

__all__ = [
    "create_range_encoder", "create_range_decoder",
    "entropy_decode_channel", "entropy_decode_finalize", "entropy_decode_index",
    "entropy_encode_channel", "entropy_encode_finalize", "entropy_encode_index",
    "pmf_to_quantized_cdf", "range_encode", "range_decode",
    "unbounded_index_range_encode", "unbounded_index_range_decode",
    "stochastic_round",
    "set_default_mode", "get_default_mode",
    "entropy_encode_finalize_device", "entropy_encode_status",
    "entropy_decode_finalize_device", "entropy_decode_status",
    "entropy_encode_channel_many", "entropy_decode_channel_many",
    "create_range_encoders", "create_range_decoders",
    "entropy_encode_finalize_device_many", "entropy_decode_finalize_device_many",
    "device_strings", "fetch_strings", "invalidate_table_cache",
]

_MODES = {None: 0, "auto": 0, "latency": 1, "throughput": 2}


def _mode_code(mode) -> int:
    if mode not in _MODES:
        raise ValueError(f"mode must be one of 'auto', 'latency', 'throughput': {mode!r}")
    return _MODES[mode]

_DTYPE_CODE = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


def _dev_i32(t, device):
    t = torch.as_tensor(t)
    if t.dtype != torch.int32:
        raise TypeError(f"expected int32 tensor, got {t.dtype}")
    return t.to(device).contiguous()


class _Tables:
    """Owns a tfc_tables*; cached per lookup tensor version."""

    def __init__(self, lookup: torch.Tensor):
        host = lookup.detach().to("cpu", torch.int32).contiguous().numpy()
        if host.ndim not in (1, 2):
            raise ValueError(f"`lookup` must be rank 1 or 2: {tuple(host.shape)}")
        rows, cols = (1, host.shape[0]) if host.ndim == 1 else host.shape
        out = C.c_void_p()
        _lib.check(_lib.lib().tfc_tables_create(
            host.ctypes.data, host.ndim, rows, cols, _lib.stream_ptr(), C.byref(out)))
        self.ptr = out
        self.count = int(_lib.lib().tfc_tables_count(out))

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                _lib.lib().tfc_tables_destroy(self.ptr)
        except Exception:
            pass


_TABLE_CACHE: "OrderedDict[tuple, tuple]" = OrderedDict()
_TABLE_CACHE_SIZE = 32


_TABLE_CACHE_LOCK = threading.Lock()      # handles may be created from several host threads


def invalidate_table_cache() -> None:
    """Drops every cached device copy of range-coding tables.  The cache is keyed on the lookup tensor's
    storage and version counter; a write through `.data` does not advance the counter — call this after one."""
    with _TABLE_CACHE_LOCK:
        _TABLE_CACHE.clear()


def _tables_for(lookup) -> _Tables:
    lookup = torch.as_tensor(lookup)
    if lookup.dtype != torch.int32:
        raise TypeError(f"`lookup` must be int32, got {lookup.dtype}")
    try:
        version = lookup._version
    except RuntimeError:            # inference tensors carry no version counter: never cached
        return _Tables(lookup)
    key = (lookup.data_ptr(), version, tuple(lookup.shape), str(lookup.device))
    with _TABLE_CACHE_LOCK:
        hit = _TABLE_CACHE.get(key)
        if hit is not None and hit[0] is lookup:
            _TABLE_CACHE.move_to_end(key)
            return hit[1]
        tables = _Tables(lookup)
        _TABLE_CACHE[key] = (lookup, tables)   # holding `lookup` pins data_ptr
        while len(_TABLE_CACHE) > _TABLE_CACHE_SIZE:
            _TABLE_CACHE.popitem(last=False)
        return tables


class EncoderHandle:
    """Stands in for the DT_VARIANT handle tensor of CreateRangeEncoder
    (cc/kernels/range_coder_kernels.cc:62-78, 484-507)."""

    def __init__(self, shape, tables: _Tables, device, mode=None, deferred_errors=False, ptr=None):
        self.shape = tuple(int(s) for s in shape)
        self.tables = tables
        self.device = device
        self.streams = int(np.prod(self.shape, dtype=np.int64))
        out = ptr
        if out is None:
            out = C.c_void_p()
            _lib.check(_lib.lib().tfc_encoder_create(tables.ptr, self.streams, _lib.stream_ptr(),
                                                    C.byref(out)))
        self.ptr = out
        if _mode_code(mode):
            _lib.check(_lib.lib().tfc_encoder_set_mode(out, _mode_code(mode)))
        if deferred_errors:
            _lib.check(_lib.lib().tfc_encoder_set_deferred_errors(out, 1))
        self.mode = mode
        self.deferred = bool(deferred_errors)
        self._keep = []       # inputs of in-flight kernels
        # deferred handles: how to issue every encode call again (one closure per call, taking a handle) — see
        # _retry_outgrown
        self._replay = []
        # native encoders this handle has been (see _retry_outgrown): decoders created on their device strings and
        # device_strings views read their blobs in place, so they live as long as the handle does
        self._retired = []
        self.blob = None      # after finalize: device uint8 [total]
        self.offsets = None   # after finalize: device int64 [streams + 1]

    def record(self, call):
        """`call(handle)` issues an encode call of this handle again: kept for deferred handles, whose slab is sized
        without looking at the data."""
        if self.deferred:
            self._replay.append(call)

    def __del__(self):
        try:
            for old in getattr(self, "_retired", ()):
                _lib.lib().tfc_encoder_destroy(old)
            if getattr(self, "ptr", None):
                _lib.lib().tfc_encoder_destroy(self.ptr)
        except Exception:
            pass


_OUTGROWN = "outgrew its output slab"


def _retry_outgrown(handle: EncoderHandle) -> None:
    """A deferred handle codes into a slab sized from the geometry alone (2 bytes per symbol and a margin); a stream of
    mostly long escape codes can need more, which the kernels notice (they never write past the slab) and the handle
    reports at its first synchronising call.  The reference never fails on encodable input
    (cc/kernels/range_coder_kernels.cc:191-322 grows a std::string), so that report is not passed on: every encode call
    of the handle is issued again on a fresh SYNCHRONISING encoder — whose calls size their slabs from a counting pass,
    or repeat with the bound no stream can exceed — and the handle continues as that encoder.  (A decoder that was
    created on the handle's device strings before the report was read decoded the incomplete strings: its
    EntropyDecodeFinalize flags say so.)  The encoder the handle was before is NOT destroyed here: decoders made by
    create_range_decoders and device_strings views read its blob in place and only keep this Python object alive —
    it is parked on the handle and freed with it.

    The replayed calls read the caller's tensors again, at retry time: inputs of a deferred handle must stay
    unmodified until its first synchronising call (fetch_strings / entropy_encode_status / entropy_encode_finalize),
    which also drops the handle's references to them."""
    fresh = EncoderHandle(handle.shape, handle.tables, handle.device, handle.mode, deferred_errors=False)
    for call in handle._replay:
        call(fresh)
    old = handle.ptr
    handle.ptr, fresh.ptr = fresh.ptr, None
    handle._keep += fresh._keep
    handle.deferred = False
    handle._replay = []
    handle.blob = handle.offsets = None
    handle.retried = True
    handle._retired.append(old)
    if getattr(handle, "finalized_on_device", False):
        # the state the caller left it in: finalized, strings in HBM (device_strings views have to be taken again)
        _lib.check(_lib.lib().tfc_encoder_finalize_device(handle.ptr, _lib.stream_ptr()))


def _with_retry(handle: EncoderHandle, fn):
    """fn() on the handle; an outgrown speculative slab is repaired once (see _retry_outgrown) and fn() repeated."""
    try:
        return fn()
    except ValueError as e:
        if _OUTGROWN not in str(e) or not handle.deferred or not handle._replay:
            raise
    _retry_outgrown(handle)
    return fn()


class DecoderHandle:
    """Stands in for the handle of CreateRangeDecoder
    (cc/kernels/range_coder_kernels.cc:80-96, 597-619)."""

    def __init__(self, shape, tables: _Tables, device, ptr):
        self.shape = tuple(int(s) for s in shape)
        self.tables = tables
        self.device = device
        self.streams = int(np.prod(self.shape, dtype=np.int64))
        self.ptr = ptr
        self._keep = []

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                _lib.lib().tfc_decoder_destroy(self.ptr)
        except Exception:
            pass


def _shape_list(shape):
    if isinstance(shape, torch.Tensor):
        shape = shape.detach().cpu().tolist()
    shape = [int(s) for s in np.asarray(shape).reshape(-1)]
    if any(s < 0 for s in shape):
        raise ValueError(f"invalid shape {shape}")
    return shape


def create_range_encoder(shape, lookup, mode=None, deferred_errors=False) -> EncoderHandle:
    """CreateRangeEncoder(shape, lookup) -> handle.

    `mode` ('auto' | 'latency' | 'throughput', include/tfc_hip.h TFC_MODE_*) picks the kernel
    family — same bytes either way; `deferred_errors` makes throughput-mode encode calls fully
    asynchronous (range errors surface at finalize / entropy_encode_status)."""
    device = _lib.require_device()
    return EncoderHandle(_shape_list(shape), _tables_for(lookup), device, mode, deferred_errors)


def create_range_encoders(n, shape, lookup, mode=None, deferred_errors=False):
    """n independent CreateRangeEncoder handles of the same shape with one allocation and one launch
    (tfc_encoder_create_many)."""
    device = _lib.require_device()
    tables = _tables_for(lookup)
    shape = _shape_list(shape)
    streams = int(np.prod(shape, dtype=np.int64))
    ptrs = (C.c_void_p * n)()
    _lib.check(_lib.lib().tfc_encoder_create_many(tables.ptr, streams, n, _lib.stream_ptr(), ptrs))
    return [EncoderHandle(shape, tables, device, mode, deferred_errors, ptr=C.c_void_p(p)) for p in ptrs]


def _check_prefix(handle_shape, value_shape, what="value"):
    if tuple(value_shape[:len(handle_shape)]) != tuple(handle_shape):
        raise ValueError(
            f"'{what}' shape should start with 'handle' shape: {what}.shape={list(value_shape)} "
            f"does not start with handle.shape={list(handle_shape)}")


def entropy_encode_channel(handle: EncoderHandle, value) -> EncoderHandle:
    """EntropyEncodeChannel(handle, value) -> aliased handle."""
    if handle.streams == 0:
        raise ValueError(f"`handle` is empty: handle.shape={list(handle.shape)}")
    value = _dev_i32(value, handle.device)
    _check_prefix(handle.shape, value.shape)
    elems = value.numel() // handle.streams
    handle._keep.append(value)
    handle.record(lambda h, v=value: entropy_encode_channel(h, v))
    _lib.check(_lib.lib().tfc_encoder_encode(handle.ptr, value.data_ptr(), None, elems,
                                             _lib.stream_ptr()))
    return handle


def entropy_encode_channel_many(handles, values):
    """EntropyEncodeChannel for several independent handles (same lookup, same shape) as ONE launch
    (include/tfc_hip.h tfc_encoder_encode_many): same results as calling entropy_encode_channel on each."""
    handles = list(handles)
    if not handles:
        return handles
    vals = []
    for h, v in zip(handles, values):
        if h.streams == 0:
            raise ValueError(f"`handle` is empty: handle.shape={list(h.shape)}")
        v = _dev_i32(v, h.device)
        _check_prefix(h.shape, v.shape)
        vals.append(v)
        h._keep.append(v)
        h.record(lambda hh, vv=v: entropy_encode_channel(hh, vv))
    elems = {v.numel() // h.streams for h, v in zip(handles, vals)}
    if len(elems) != 1:
        raise ValueError("entropy_encode_channel_many: all values must have the same shape")
    n = len(handles)
    hp = (C.c_void_p * n)(*[h.ptr for h in handles])
    vp = (C.c_void_p * n)(*[v.data_ptr() for v in vals])
    _lib.check(_lib.lib().tfc_encoder_encode_many(n, hp, vp, None, elems.pop(), _lib.stream_ptr()))
    return handles


def entropy_encode_index(handle: EncoderHandle, index, value) -> EncoderHandle:
    """EntropyEncodeIndex(handle, index, value) -> aliased handle."""
    if handle.streams == 0:
        raise ValueError(f"`handle` is empty: handle.shape={list(handle.shape)}")
    value = _dev_i32(value, handle.device)
    index = _dev_i32(index, handle.device)
    _check_prefix(handle.shape, value.shape)
    if index.shape != value.shape:
        raise ValueError(
            f"'index' shape should match 'value' shape: index.shape={list(index.shape)} "
            f"!= value.shape={list(value.shape)}")
    elems = value.numel() // handle.streams
    handle._keep += [value, index]
    handle.record(lambda h, i=index, v=value: entropy_encode_index(h, i, v))
    _lib.check(_lib.lib().tfc_encoder_encode(handle.ptr, value.data_ptr(), index.data_ptr(),
                                             elems, _lib.stream_ptr()))
    return handle


def _finalize_device(handle: EncoderHandle):
    total = C.c_int64()
    _with_retry(handle, lambda: _lib.check(_lib.lib().tfc_encoder_finalize(handle.ptr, _lib.stream_ptr(), C.byref(total))))
    handle._keep.clear()
    handle._replay = []
    blob = torch.empty(max(total.value, 1), dtype=torch.uint8, device=handle.device)
    offsets = torch.empty(handle.streams + 1, dtype=torch.int64, device=handle.device)
    _lib.check(_lib.lib().tfc_encoder_read(handle.ptr, blob.data_ptr(), offsets.data_ptr(), 1,
                                           _lib.stream_ptr()))
    handle.blob = blob[:total.value]
    handle.offsets = offsets
    return handle.blob, handle.offsets


def entropy_encode_finalize_device(handle: EncoderHandle) -> EncoderHandle:
    """EntropyEncodeFinalize without any host synchronisation: the packed strings stay in HBM inside
    the handle; pass the handle itself to create_range_decoder (stream-ordered), and call
    entropy_encode_status / entropy_encode_finalize when the host needs errors or bytes."""
    if handle.streams == 0:
        raise ValueError(f"`handle` is empty: {list(handle.shape)}")
    _lib.check(_lib.lib().tfc_encoder_finalize_device(handle.ptr, _lib.stream_ptr()))
    handle.finalized_on_device = True
    return handle


def entropy_encode_finalize_device_many(handles):
    """entropy_encode_finalize_device for several handles: three launches in all for handles coded by
    entropy_encode_channel_many (tfc_encoder_finalize_device_many)."""
    handles = list(handles)
    n = len(handles)
    if n:
        hp = (C.c_void_p * n)(*[h.ptr for h in handles])
        _lib.check(_lib.lib().tfc_encoder_finalize_device_many(n, hp, _lib.stream_ptr()))
        for h in handles:
            h.finalized_on_device = True
    return handles


class _DeviceSpan:
    """A span of device memory owned by a handle, exposed through __cuda_array_interface__ so that torch can
    view it without a copy (`owner` keeps the handle alive as long as the view is)."""

    def __init__(self, ptr, shape, typestr, owner):
        self.owner = owner
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def device_strings(handle: EncoderHandle):
    """Zero-copy torch views of a finalized handle's packed strings in HBM: (blob uint8 [capacity],
    offsets int64 [streams + 1]); stream s is blob[offsets[s]:offsets[s + 1]].  After
    entropy_encode_finalize_device the blob has the slabs' capacity — the total is offsets[-1], on the
    device.  The views are valid in the order of the stream the handle was finalized on."""
    blob_p, off_p = C.c_void_p(), C.c_void_p()
    _lib.check(_lib.lib().tfc_encoder_result(handle.ptr, C.byref(blob_p), C.byref(off_p)))
    cap = C.c_int64()
    _lib.check(_lib.lib().tfc_encoder_capacity(handle.ptr, C.byref(cap)))
    offsets = torch.as_tensor(_DeviceSpan(off_p.value, (handle.streams + 1,), "<i8", handle), device=handle.device)
    if cap.value <= 0 or not blob_p.value:
        return torch.empty(0, dtype=torch.uint8, device=handle.device), offsets
    blob = torch.as_tensor(_DeviceSpan(blob_p.value, (cap.value,), "|u1", handle), device=handle.device)
    return blob, offsets


def fetch_strings(handle: EncoderHandle):
    """The strings of a handle finalized with entropy_encode_finalize_device, on the host (numpy object
    array of bytes shaped like the handle).  Synchronises the current stream; raises a deferred range
    error.  Same result as entropy_encode_finalize on a handle that was not finalized yet."""
    total = entropy_encode_status(handle)
    if total < 0:
        return entropy_encode_finalize(handle)
    blob = np.empty(max(total, 1), np.uint8)
    off = np.empty(handle.streams + 1, np.int64)
    _lib.check(_lib.lib().tfc_encoder_read(handle.ptr, blob.ctypes.data, off.ctypes.data, 0, _lib.stream_ptr()))
    handle._keep.clear()
    data = blob.tobytes()
    out = np.empty(handle.streams, dtype=object)
    for i in range(handle.streams):
        out[i] = data[off[i]:off[i + 1]]
    return out.reshape(handle.shape)


def entropy_encode_status(handle: EncoderHandle) -> int:
    """Synchronises; raises a deferred range error, returns the total byte count (-1 before finalize).  A deferred
    handle whose speculative slab a stream outgrew is coded again here (_retry_outgrown): never an error."""
    total = C.c_int64()
    _with_retry(handle, lambda: _lib.check(_lib.lib().tfc_encoder_status(handle.ptr, _lib.stream_ptr(), C.byref(total))))
    # the calls are known to have fitted (or were repeated): nothing will be replayed, the inputs may go
    handle._replay = []
    return int(total.value)


def strings_from_blob(blob, offsets, shape):
    """(uint8 blob, int64 offsets) -> numpy object array of bytes with `shape`."""
    blob_h = blob.detach().cpu().numpy().tobytes()
    off = offsets.detach().cpu().numpy()
    out = np.empty(len(off) - 1, dtype=object)
    for i in range(len(off) - 1):
        out[i] = blob_h[off[i]:off[i + 1]]
    return out.reshape(shape)


def entropy_encode_finalize(handle: EncoderHandle):
    """EntropyEncodeFinalize(handle) -> encoded strings shaped like handle."""
    if handle.streams == 0:
        raise ValueError(f"`handle` is empty: {list(handle.shape)}")
    blob, offsets = _finalize_device(handle)
    return strings_from_blob(blob, offsets, handle.shape)


def blob_from_strings(strings):
    """numpy/bytes container -> (uint8 ndarray blob, int64 ndarray offsets, shape)."""
    if isinstance(strings, (bytes, bytearray)):
        arr = np.empty((), dtype=object)
        arr[()] = bytes(strings)
    else:
        arr = np.asarray(strings, dtype=object)
    flat = [bytes(s) for s in arr.reshape(-1)]
    off = np.zeros(len(flat) + 1, np.int64)
    if flat:
        off[1:] = np.cumsum([len(s) for s in flat])
    blob = np.frombuffer(b"".join(flat), np.uint8).copy() if off[-1] else np.zeros(1, np.uint8)
    return blob, off, arr.shape


def create_range_decoder(encoded, lookup, mode=None) -> DecoderHandle:
    """CreateRangeDecoder(encoded, lookup) -> handle.  `encoded` is a container
    of bytes (any shape), a (device blob, device offsets, shape) triple, or a finalized
    EncoderHandle (its device-resident strings are read in place)."""
    handle = _create_range_decoder(encoded, lookup)
    if _mode_code(mode):
        _lib.check(_lib.lib().tfc_decoder_set_mode(handle.ptr, _mode_code(mode)))
    return handle


def create_range_decoders(encoders, lookup, mode=None):
    """One CreateRangeDecoder per finalized EncoderHandle, reading its device-resident strings in place:
    one allocation, one launch (tfc_decoder_create_many)."""
    encoders = list(encoders)
    n = len(encoders)
    if n == 0:
        return []
    device = _lib.require_device()
    tables = _tables_for(lookup)
    ep = (C.c_void_p * n)(*[e.ptr for e in encoders])
    ptrs = (C.c_void_p * n)()
    _lib.check(_lib.lib().tfc_decoder_create_many(tables.ptr, n, ep, _lib.stream_ptr(), ptrs))
    out = []
    for e, p in zip(encoders, ptrs):
        d = DecoderHandle(e.shape, tables, device, C.c_void_p(p))
        d._keep.append(e)        # the decoder reads the encoder's blob in place
        if _mode_code(mode):
            _lib.check(_lib.lib().tfc_decoder_set_mode(d.ptr, _mode_code(mode)))
        out.append(d)
    return out


def _create_range_decoder(encoded, lookup) -> DecoderHandle:
    device = _lib.require_device()
    tables = _tables_for(lookup)
    out = C.c_void_p()
    if isinstance(encoded, EncoderHandle):
        blob_p, off_p = C.c_void_p(), C.c_void_p()
        _lib.check(_lib.lib().tfc_encoder_result(encoded.ptr, C.byref(blob_p), C.byref(off_p)))
        _lib.check(_lib.lib().tfc_decoder_create(tables.ptr, blob_p, off_p, encoded.streams, 1,
                                                _lib.stream_ptr(), C.byref(out)))
        handle = DecoderHandle(encoded.shape, tables, device, out)
        handle._keep.append(encoded)        # the decoder reads the encoder's blob in place
        return handle
    if isinstance(encoded, tuple) and len(encoded) == 3 and isinstance(encoded[0], torch.Tensor):
        blob, offsets, shape = encoded
        shape = tuple(int(s) for s in shape)
        streams = int(np.prod(shape, dtype=np.int64))
        if streams == 0:
            raise ValueError(f"`encoded` is empty: {list(shape)}")
        blob = blob.to(device).contiguous()
        offsets = offsets.to(device, torch.int64).contiguous()
        _lib.check(_lib.lib().tfc_decoder_create(tables.ptr, blob.data_ptr(), offsets.data_ptr(),
                                                streams, 1, _lib.stream_ptr(), C.byref(out)))
        handle = DecoderHandle(shape, tables, device, out)
        handle._keep += [blob, offsets]     # the decoder reads them in place (borrowed)
        return handle
    else:
        blob, off, shape = blob_from_strings(encoded)
        streams = len(off) - 1
        if streams == 0:
            raise ValueError(f"`encoded` is empty: {list(shape)}")
        _lib.check(_lib.lib().tfc_decoder_create(tables.ptr, blob.ctypes.data, off.ctypes.data,
                                                streams, 0, _lib.stream_ptr(), C.byref(out)))
    return DecoderHandle(shape, tables, device, out)


def _decode(handle: DecoderHandle, index, shape, Tdecoded):
    if Tdecoded not in (torch.int32, None):
        raise TypeError("Tdecoded must be int32")
    if handle.streams == 0:
        raise ValueError(f"`handle` is empty: {list(handle.shape)}")
    suffix = _shape_list(shape)
    out_shape = tuple(handle.shape) + tuple(suffix)
    elems = int(np.prod(suffix, dtype=np.int64))
    out = torch.empty(out_shape, dtype=torch.int32, device=handle.device)
    iptr = None
    if index is not None:
        index = _dev_i32(index, handle.device)
        if tuple(index.shape) != out_shape:
            raise ValueError(
                "'index' shape should match 'handle' shape + 'shape': "
                f"index.shape={list(index.shape)}, handle.shape={list(handle.shape)}, "
                f"shape={suffix}")
        handle._keep.append(index)
        iptr = index.data_ptr()
    _lib.check(_lib.lib().tfc_decoder_decode(handle.ptr, iptr, out.data_ptr(), elems,
                                             _lib.stream_ptr()))
    return handle, out


def entropy_decode_channel(handle: DecoderHandle, shape, Tdecoded=torch.int32):
    """EntropyDecodeChannel(handle, shape, Tdecoded) -> (aliased handle, decoded)."""
    return _decode(handle, None, shape, Tdecoded)


def entropy_decode_channel_many(handles, shape, Tdecoded=torch.int32):
    """EntropyDecodeChannel for several independent handles as ONE launch
    (tfc_decoder_decode_many) -> (handles, list of decoded tensors)."""
    if Tdecoded not in (torch.int32, None):
        raise TypeError("Tdecoded must be int32")
    handles = list(handles)
    if not handles:
        return handles, []
    suffix = _shape_list(shape)
    elems = int(np.prod(suffix, dtype=np.int64))
    outs = []
    for h in handles:
        if h.streams == 0:
            raise ValueError(f"`handle` is empty: {list(h.shape)}")
        outs.append(torch.empty(tuple(h.shape) + tuple(suffix), dtype=torch.int32, device=h.device))
    n = len(handles)
    hp = (C.c_void_p * n)(*[h.ptr for h in handles])
    op = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
    _lib.check(_lib.lib().tfc_decoder_decode_many(n, hp, None, op, elems, _lib.stream_ptr()))
    return handles, outs


def entropy_decode_index(handle: DecoderHandle, index, shape, Tdecoded=torch.int32):
    """EntropyDecodeIndex(handle, index, shape, Tdecoded) -> (aliased handle, decoded)."""
    return _decode(handle, index, shape, Tdecoded)


def entropy_decode_finalize(handle: DecoderHandle) -> torch.Tensor:
    """EntropyDecodeFinalize(handle) -> bool tensor shaped like handle."""
    if handle.streams == 0:
        raise ValueError(f"`handle` is empty: {list(handle.shape)}")
    ok = np.zeros(handle.streams, np.uint8)
    _lib.check(_lib.lib().tfc_decoder_finalize(handle.ptr, ok.ctypes.data, _lib.stream_ptr()))
    handle._keep.clear()
    return torch.from_numpy(ok.astype(bool)).reshape(handle.shape)


def entropy_decode_finalize_device(handle: DecoderHandle) -> torch.Tensor:
    """EntropyDecodeFinalize without host synchronisation -> uint8 device tensor shaped like handle
    (1 = the weak end-of-stream check passed); entropy_decode_status reports a deferred index error."""
    if handle.streams == 0:
        raise ValueError(f"`handle` is empty: {list(handle.shape)}")
    ok = torch.empty(handle.streams, dtype=torch.uint8, device=handle.device)
    _lib.check(_lib.lib().tfc_decoder_finalize_device(handle.ptr, ok.data_ptr(), _lib.stream_ptr()))
    return ok.reshape(handle.shape)


def entropy_decode_finalize_device_many(handles) -> torch.Tensor:
    """entropy_decode_finalize_device for several handles of one shape -> uint8 device tensor
    [len(handles), *shape], one launch for handles that came from create_range_decoders."""
    handles = list(handles)
    n = len(handles)
    ok = torch.empty((n,) + tuple(handles[0].shape), dtype=torch.uint8, device=handles[0].device)
    hp = (C.c_void_p * n)(*[h.ptr for h in handles])
    _lib.check(_lib.lib().tfc_decoder_finalize_device_many(n, hp, ok.data_ptr(), _lib.stream_ptr()))
    return ok


def entropy_decode_status(handle: DecoderHandle) -> None:
    _lib.check(_lib.lib().tfc_decoder_status(handle.ptr, _lib.stream_ptr()))
    handle._keep.clear()


def pmf_to_quantized_cdf(pmf, precision: int) -> torch.Tensor:
    """PmfToQuantizedCdf(pmf; precision) -> cdf with last dim + 1."""
    precision = int(precision)
    if not 0 < precision <= 16:
        raise ValueError(f"`precision` must be in [1, 16]: {precision}")
    device = _lib.require_device()
    pmf = torch.as_tensor(pmf)
    if pmf.dtype != torch.float32:
        raise TypeError(f"`pmf` must be float32, got {pmf.dtype}")
    if pmf.dim() < 1:
        raise ValueError("`pmf` should be at least 1-D.")
    n = pmf.shape[-1]
    if n <= 1:
        raise ValueError("`pmf` size should be at least 2 in the last axis.")
    pmf = pmf.to(device).contiguous()
    if not bool((torch.isfinite(pmf) & (pmf >= 0)).all()):
        bad = pmf[~(torch.isfinite(pmf) & (pmf >= 0))][0].item()
        raise ValueError(
            f"`pmf` has non-finite or negative element: {bad}. Please check for numerical "
            "problems in the probability computation.")
    rows = pmf.numel() // n
    cdf = torch.empty(pmf.shape[:-1] + (n + 1,), dtype=torch.int32, device=device)
    _lib.check(_lib.lib().tfc_pmf_to_quantized_cdf(pmf.data_ptr(), rows, n, precision,
                                                   cdf.data_ptr(), _lib.stream_ptr()))
    return cdf


def _check_precision_debug(precision, debug_level):
    if not 0 < int(precision) <= 16:
        raise ValueError(f"`precision` must be in [1, 16]: {precision}")
    if int(debug_level) not in (0, 1):
        raise ValueError(f"`debug_level` must be 0 or 1: {debug_level}")


def range_encode(data, cdf, precision: int, debug_level: int = 1) -> bytes:
    """RangeEncode(data:int16, cdf:int32; precision, debug_level) -> bytes."""
    _check_precision_debug(precision, debug_level)
    device = _lib.require_device()
    data = torch.as_tensor(data)
    cdf = torch.as_tensor(cdf)
    if data.dtype != torch.int16 or cdf.dtype != torch.int32:
        raise TypeError("range_encode expects int16 data and int32 cdf")
    data = data.to(device).contiguous()
    cdf = cdf.to(device).contiguous()
    ds = np.array(data.shape, np.int64)
    cs = np.array(cdf.shape, np.int64)
    out = C.c_void_p()
    n = C.c_int64()
    _lib.check(_lib.lib().tfc_range_encode(
        data.data_ptr(), ds.ctypes.data, data.dim(), cdf.data_ptr(), cs.ctypes.data, cdf.dim(),
        int(precision), int(debug_level), _lib.stream_ptr(), C.byref(out), C.byref(n)))
    try:
        return C.string_at(out, n.value)
    finally:
        _lib.lib().tfc_free(out)


def range_decode(encoded, shape, cdf, precision: int, debug_level: int = 1) -> torch.Tensor:
    """RangeDecode(encoded, shape, cdf; precision, debug_level) -> int16 tensor."""
    _check_precision_debug(precision, debug_level)
    device = _lib.require_device()
    if isinstance(encoded, np.ndarray):
        if encoded.shape != ():
            raise ValueError(f"Invalid `encoded` shape: {list(encoded.shape)}")
        encoded = encoded[()]
    if not isinstance(encoded, (bytes, bytearray)):
        raise ValueError("Invalid `encoded` shape: expected a scalar byte string")
    if isinstance(shape, torch.Tensor) and shape.dim() != 1:
        raise ValueError(f"Invalid `shape` shape: {list(shape.shape)}")
    shape = _shape_list(shape)
    cdf = torch.as_tensor(cdf)
    if cdf.dtype != torch.int32:
        raise TypeError("range_decode expects int32 cdf")
    cdf = cdf.to(device).contiguous()
    out = torch.empty(shape, dtype=torch.int16, device=device)
    os_ = np.array(shape, np.int64)
    cs = np.array(cdf.shape, np.int64)
    buf = np.frombuffer(bytes(encoded), np.uint8).copy() if len(encoded) else np.zeros(1, np.uint8)
    _lib.check(_lib.lib().tfc_range_decode(
        buf.ctypes.data, len(encoded), os_.ctypes.data, len(shape), cdf.data_ptr(),
        cs.ctypes.data, cdf.dim(), int(precision), int(debug_level), _lib.stream_ptr(),
        out.data_ptr()))
    return out


def _unbounded_args(index, cdf, cdf_size, offset, device):
    """Shape checks of CheckArgumentShapes (unbounded_index_range_coding_kernels.cc:115-143)."""
    index, cdf = _dev_i32(index, device), _dev_i32(cdf, device)
    cdf_size, offset = _dev_i32(cdf_size, device), _dev_i32(offset, device)
    if cdf.dim() != 2 or cdf.shape[1] < 3:
        raise ValueError(f"'cdf' should be 2-D and cdf.dim_size(1) >= 3: {list(cdf.shape)}")
    if cdf_size.dim() != 1 or cdf_size.shape[0] != cdf.shape[0]:
        raise ValueError("'cdf_size' should be 1-D and its length should match the number of rows in 'cdf': "
                         f"{list(cdf_size.shape)}")
    if offset.dim() != 1 or offset.shape[0] != cdf.shape[0]:
        raise ValueError("'offset' should be 1-D and its length should match the number of rows in 'cdf': "
                         f"offset.shape={list(offset.shape)}, cdf.shape={list(cdf.shape)}")
    return index, cdf, cdf_size, offset


def unbounded_index_range_encode(data, index, cdf, cdf_size, offset, precision: int, overflow_width: int,
                                 debug_level: int = 1) -> bytes:
    """UnboundedIndexRangeEncode (deprecated op, cc/ops/range_coding_ops.cc / kernels
    unbounded_index_range_coding_kernels.cc:146-249) -> one byte string for the whole tensor."""
    device = _lib.require_device()
    index, cdf, cdf_size, offset = _unbounded_args(index, cdf, cdf_size, offset, device)
    data = _dev_i32(data, device)
    if data.shape != index.shape:
        raise ValueError(f"`data` and `index` should have the same shape: data.shape={list(data.shape)}, "
                         f"index.shape={list(index.shape)}")
    out, n = C.c_void_p(), C.c_int64()
    _lib.check(_lib.lib().tfc_unbounded_index_range_encode(
        data.data_ptr(), index.data_ptr(), data.numel(), cdf.data_ptr(), cdf.shape[0], cdf.shape[1],
        cdf_size.data_ptr(), offset.data_ptr(), int(precision), int(overflow_width), int(debug_level),
        _lib.stream_ptr(), C.byref(out), C.byref(n)))
    try:
        return C.string_at(out.value, n.value) if n.value else b""
    finally:
        _lib.lib().tfc_free(out)


def unbounded_index_range_decode(encoded, index, cdf, cdf_size, offset, precision: int, overflow_width: int,
                                 debug_level: int = 1) -> torch.Tensor:
    """UnboundedIndexRangeDecode -> int32 tensor shaped like `index`
    (unbounded_index_range_coding_kernels.cc:259-367)."""
    device = _lib.require_device()
    if isinstance(encoded, np.ndarray):
        if encoded.shape != ():
            raise ValueError(f"`encoded` should be a scalar: {list(encoded.shape)}")
        encoded = encoded[()]
    if not isinstance(encoded, (bytes, bytearray)):
        raise ValueError("`encoded` should be a scalar: expected a byte string")
    index, cdf, cdf_size, offset = _unbounded_args(index, cdf, cdf_size, offset, device)
    out = torch.empty(index.shape, dtype=torch.int32, device=device)
    buf = np.frombuffer(bytes(encoded), np.uint8).copy() if len(encoded) else np.zeros(1, np.uint8)
    _lib.check(_lib.lib().tfc_unbounded_index_range_decode(
        buf.ctypes.data, len(encoded), index.data_ptr(), index.numel(), cdf.data_ptr(), cdf.shape[0],
        cdf.shape[1], cdf_size.data_ptr(), offset.data_ptr(), int(precision), int(overflow_width),
        int(debug_level), out.data_ptr(), _lib.stream_ptr()))
    return out


def stochastic_round(inputs, step_size, seed) -> torch.Tensor:
    """StochasticRound (cc/ops/quantization_ops.cc:21-44, quantization_kernels.cc:47-96): rounds
    `inputs / step_size` down or up with probability equal to the fractional part; int32, same shape.

    `seed`: int32 values of any shape — equal seeds give the reference's results bit for bit (one
    xoshiro256+ stream in flat element order); an empty seed seeds from the clock."""
    device = _lib.require_device()
    inputs = torch.as_tensor(inputs)
    if inputs.dtype not in _DTYPE_CODE:
        raise TypeError(f"`inputs` must be bfloat16, float16 or float32: {inputs.dtype}")
    step = torch.as_tensor(step_size)
    if step.dim() != 0:
        raise ValueError("step_size must be a scalar.")
    seed = np.ascontiguousarray(
        seed.detach().cpu().numpy() if isinstance(seed, torch.Tensor) else np.asarray(seed, dtype=np.int64))
    if seed.size and (seed.min() < -2**31 or seed.max() > 2**31 - 1):
        raise TypeError("`seed` must hold int32 values")
    seed = seed.astype(np.int32).reshape(-1)
    inputs = inputs.to(device).contiguous()
    out = torch.empty(inputs.shape, dtype=torch.int32, device=device)
    _lib.check(_lib.lib().tfc_stochastic_round(
        inputs.data_ptr(), _DTYPE_CODE[inputs.dtype], inputs.numel(), float(step),
        seed.ctypes.data if seed.size else None, seed.size, out.data_ptr(), _lib.stream_ptr()))
    return out


def set_default_mode(mode) -> None:
    """Process-wide default kernel family for handles created without `mode`
    (include/tfc_hip.h, tfc_set_default_mode)."""
    _lib.check(_lib.lib().tfc_set_default_mode(_mode_code(mode)))


def get_default_mode() -> str:
    return ("auto", "latency", "throughput")[_lib.lib().tfc_get_default_mode()]
