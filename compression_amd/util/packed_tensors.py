"""PackedTensors — the `.tfci` container of the reference (python/util/packed_tensors.py:25-100).

The reference stores the compressed strings and shape vectors in a serialized
`tf.train.Example`: features keyed chr(1), chr(2), ... in pack order, each a
`bytes_list` (strings), `int64_list` (integer vectors) or `float_list`, plus an
optional "MD" bytes feature naming the model.  TensorFlow is not a dependency
here, so the protobuf wire format of those four messages is read and written
directly (proto3, packed repeated scalars):

    Example   { Features features = 1; }
    Features  { map<string, Feature> feature = 1; }     entry: key = 1, value = 2
    Feature   { oneof { BytesList bytes_list = 1; FloatList float_list = 2; Int64List int64_list = 3; } }
    BytesList { repeated bytes value = 1; }   FloatList { repeated float value = 1 [packed]; }
    Int64List { repeated int64 value = 1 [packed]; }

Map entries are written sorted by key (what protobuf's deterministic mode
emits); any order — and unpacked scalars — are accepted when reading, so files
written by the reference parse, and files written here parse there.
"""
from __future__ import annotations

import struct

import numpy as np
import torch

__all__ = ["PackedTensors"]


def _varint(v: int) -> bytes:
    v &= (1 << 64) - 1                      # int64 -> two's complement, 10 bytes when negative
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _read_varint(buf: bytes, pos: int):
    shift = result = 0
    while True:
        if pos >= len(buf):
            raise ValueError("truncated varint")
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError("varint too long")


def _len_delimited(field: int, payload: bytes) -> bytes:
    return _varint(field << 3 | 2) + _varint(len(payload)) + payload


def _fields(buf: bytes):
    """Yields (field number, wire type, value) of one message."""
    pos = 0
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        field, wire = key >> 3, key & 7
        if wire == 0:
            val, pos = _read_varint(buf, pos)
        elif wire == 1:
            val, pos = buf[pos:pos + 8], pos + 8
        elif wire == 2:
            n, pos = _read_varint(buf, pos)
            if pos + n > len(buf):
                raise ValueError("truncated field")
            val, pos = buf[pos:pos + n], pos + n
        elif wire == 5:
            val, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError(f"unsupported wire type {wire}")
        yield field, wire, val


def _signed64(v: int) -> int:
    return v - (1 << 64) if v >= 1 << 63 else v


class _Feature:
    """kind in {"bytes", "float", "int64", None}; values a list."""

    def __init__(self, kind=None, values=None):
        self.kind, self.values = kind, list(values or [])

    def serialize(self) -> bytes:
        if self.kind == "bytes":
            inner = b"".join(_len_delimited(1, bytes(v)) for v in self.values)
            return _len_delimited(1, inner)
        if self.kind == "float":
            inner = _len_delimited(1, struct.pack(f"<{len(self.values)}f", *self.values)) if self.values else b""
            return _len_delimited(2, inner)
        if self.kind == "int64":
            inner = _len_delimited(1, b"".join(_varint(int(v)) for v in self.values)) if self.values else b""
            return _len_delimited(3, inner)
        return b""

    @classmethod
    def parse(cls, buf: bytes) -> "_Feature":
        feat = cls()
        for field, wire, val in _fields(buf):
            if wire != 2 or field not in (1, 2, 3):
                continue
            feat.kind, feat.values = {1: "bytes", 2: "float", 3: "int64"}[field], []
            for f2, w2, v2 in _fields(val):
                if f2 != 1:
                    continue
                if field == 1:
                    feat.values.append(bytes(v2))
                elif field == 2:
                    if w2 == 2:
                        feat.values.extend(struct.unpack(f"<{len(v2) // 4}f", v2))
                    else:
                        feat.values.append(struct.unpack("<f", v2)[0])
                else:
                    if w2 == 2:
                        pos = 0
                        while pos < len(v2):
                            x, pos = _read_varint(v2, pos)
                            feat.values.append(_signed64(x))
                    else:
                        feat.values.append(_signed64(v2))
        return feat


class PackedTensors:
    """Packed representation of compressed tensors (python/util/packed_tensors.py:25-100):
    several rank-1 integer / float / string tensors in one byte string, plus an
    optional model identifier."""

    def __init__(self, string=None):
        self._features: dict[str, _Feature] = {}
        if string:
            self.string = string

    # -- model identifier ("MD" feature, :41-53) --------------------------------
    @property
    def model(self):
        feat = self._features.get("MD")
        if feat is None or not feat.values:
            raise IndexError("no model identifier stored")
        return feat.values[0].decode("ascii")

    @model.setter
    def model(self, value):
        self._features["MD"] = _Feature("bytes", [value.encode("ascii")])

    @model.deleter
    def model(self):
        self._features.pop("MD", None)

    # -- serialisation (:55-62) -----------------------------------------------------
    @property
    def string(self) -> bytes:
        entries = b"".join(
            _len_delimited(1, _len_delimited(1, key.encode("utf-8")) + _len_delimited(2, feat.serialize()))
            for key, feat in sorted(self._features.items(), key=lambda kv: kv[0].encode("utf-8")))
        return _len_delimited(1, entries) if self._features else b""

    @string.setter
    def string(self, value):
        self._features = {}
        for field, wire, features in _fields(bytes(value)):
            if field != 1 or wire != 2:
                continue
            for f2, w2, entry in _fields(features):
                if f2 != 1 or w2 != 2:
                    continue
                key, feat = "", _Feature()
                for f3, w3, v3 in _fields(entry):
                    if f3 == 1 and w3 == 2:
                        key = v3.decode("utf-8")
                    elif f3 == 2 and w3 == 2:
                        feat = _Feature.parse(v3)
                self._features[key] = feat

    # -- pack / unpack (:64-100) ----------------------------------------------------
    def pack(self, tensors):
        """Packs rank-1 tensors / arrays / sequences: integers -> int64_list, floats ->
        float_list, bytes -> bytes_list; features are named chr(1), chr(2), ..."""
        i = 1
        for tensor in tensors:
            if isinstance(tensor, torch.Tensor):
                tensor = tensor.detach().cpu().numpy()
            arr = np.asarray(tensor)
            if arr.dtype == object or arr.dtype.kind in "SU":
                if arr.ndim != 1:
                    raise RuntimeError(f"Unexpected tensor rank: {arr.ndim}.")
                feat = _Feature("bytes", [v if isinstance(v, (bytes, bytearray)) else str(v).encode()
                                          for v in arr])
            else:
                if arr.ndim != 1:
                    raise RuntimeError(f"Unexpected tensor rank: {arr.ndim}.")
                if arr.dtype.kind in "iub":
                    feat = _Feature("int64", [int(v) for v in arr])
                elif arr.dtype.kind == "f":
                    feat = _Feature("float", [float(v) for v in arr.astype(np.float32)])
                else:
                    raise RuntimeError(f"Unexpected tensor dtype: '{arr.dtype}'.")
            self._features[chr(i)] = feat
            i += 1
        while chr(i) in self._features:       # delete any remaining, previously set arrays
            del self._features[chr(i)]
            i += 1

    def unpack(self, dtypes):
        """Unpacks values based on dtypes: numpy / torch integer or float dtypes give numpy
        arrays of that dtype, `bytes` (or "string") gives a numpy object array of bytes."""
        out = []
        for i, dtype in enumerate(dtypes):
            feat = self._features.get(chr(i + 1), _Feature())
            if dtype in (bytes, str, "string", object, np.object_):
                arr = np.empty(len(feat.values) if feat.kind == "bytes" else 0, dtype=object)
                if feat.kind == "bytes":
                    arr[:] = feat.values
                out.append(arr)
                continue
            if isinstance(dtype, torch.dtype):
                dtype = torch.empty(0, dtype=dtype).numpy().dtype
            dtype = np.dtype(dtype)
            if dtype.kind in "iu":
                out.append(np.asarray(feat.values if feat.kind == "int64" else [], dtype=dtype))
            elif dtype.kind == "f":
                out.append(np.asarray(feat.values if feat.kind == "float" else [], dtype=dtype))
            else:
                raise RuntimeError(f"Unexpected dtype: '{dtype}'.")
        return out
