"""Stream plumbing for model pipelines on one GPU: the coder and the transforms on disjoint
compute units, several batches in flight.

Why: a model step is transforms (SignalConv2D / GDN: want every CU for tens of milliseconds) plus
range coding (one wave per image and SIMD: a handful of CUs for tens of milliseconds, bound by the
length of one stream's chain whatever the batch).  Run one behind the other the step costs their
sum; run on the same CUs at the same time each slows the other down (profiles/r02_notes.md).  Here
the coder gets its own CUs through a CU-masked HIP stream (include/tfc_hip.h,
tfc_stream_create_cu_mask), the transforms get the rest, and `depth` batches are in flight so that
batch k's coding overlaps batch k + 1's transforms: the step tends to max(transforms, coder).

The reference has no counterpart — its ops share TensorFlow's inter/intra-op thread pools
(range_coder_kernels.cc:212-215) — but the structure mirrors what its executor does with
independent ops of a graph.

Nothing here synchronises with the host; a `Lane` only orders its two streams with events.
"""
from __future__ import annotations

import contextlib
import ctypes as C

import numpy as np
import torch

from . import _lib

__all__ = ["CoderPartition", "Lane", "inline_lane"]


class Lane:
    """One batch in flight: a transform stream and a coder stream.  `with lane.on("coder"):` makes
    the coder stream current, ordered behind everything the lane enqueued before (events between the
    two streams are only placed where the stream changes)."""

    def __init__(self, transform, coder):
        self.transform, self.coder = transform, coder
        self._last_stream = None
        self._last_event = None

    def begin(self, after=None):
        """Orders the lane behind `after` (a stream; default: the current one) — the producer of the
        batch it is about to process."""
        after = after or torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(after)
        self._last_stream, self._last_event = after, ev
        return self

    @contextlib.contextmanager
    def on(self, which):
        s = self.transform if which == "transform" else self.coder
        if s is None:                       # inline lane: whatever stream is current
            yield
            return
        if self._last_event is not None and self._last_stream is not s:
            s.wait_event(self._last_event)
        with torch.cuda.stream(s):
            yield
        ev = torch.cuda.Event()
        ev.record(s)
        self._last_stream, self._last_event = s, ev

    def end_event(self):
        """Event behind everything the lane has enqueued so far (None for an inline lane)."""
        return self._last_event

    def join(self, stream=None):
        """Makes `stream` (default: the current one) wait for the lane."""
        if self._last_event is not None:
            (stream or torch.cuda.current_stream()).wait_event(self._last_event)


def inline_lane():
    """A lane whose stages all run on the caller's current stream (no partition)."""
    return Lane(None, None)


class CoderPartition:
    """`depth` lanes whose coder streams are restricted to `coder_cus` compute units and whose
    transform streams get the others.  Consecutive bits of a HIP CU mask alternate over the XCDs, so
    the coder's share is spread evenly over the eight dies (and their L2s)."""

    def __init__(self, coder_cus=32, depth=2, device=None, mode="masked"):
        """mode: "masked" (disjoint CU sets), "plain" (two ordinary streams per lane: the hardware shares
        the CUs), "single" (one ordinary stream per lane: only whole steps overlap)."""
        _lib.require_device()
        self.mode = mode
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        n = C.c_int()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().tfc_device_compute_units(C.byref(n)))
            self.total_cus = int(n.value)
            self.coder_cus = int(max(0, min(coder_cus, self.total_cus - 8)))
            words = (self.total_cus + 31) // 32
            self._raw = []
            self.lanes = []
            for _ in range(int(depth)):
                if self.coder_cus == 0 or mode == "single":
                    t = c = torch.cuda.Stream(device=self.device)
                elif mode == "plain":
                    t, c = torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device, priority=-1)
                else:
                    coder_bits = (1 << self.coder_cus) - 1
                    rest_bits = ((1 << self.total_cus) - 1) & ~coder_bits
                    coder_mask = [(coder_bits >> (32 * w)) & 0xFFFFFFFF for w in range(words)]
                    rest_mask = [(rest_bits >> (32 * w)) & 0xFFFFFFFF for w in range(words)]
                    c = torch.cuda.Stream(device=self.device) if mode == "transform-masked" else self._masked(coder_mask)
                    t = torch.cuda.Stream(device=self.device) if mode == "coder-masked" else self._masked(rest_mask)
                self.lanes.append(Lane(t, c))

    def _masked(self, mask):
        mask = np.asarray(mask, np.uint32)
        out = C.c_void_p()
        _lib.check(_lib.lib().tfc_stream_create_cu_mask(mask.ctypes.data, int(mask.size), C.byref(out)))
        self._raw.append(out)
        return torch.cuda.ExternalStream(out.value, device=self.device)

    def lane(self, k):
        return self.lanes[k % len(self.lanes)]

    def synchronize(self):
        for lane in self.lanes:
            for s in {lane.transform, lane.coder}:
                s.synchronize()

    def close(self):
        self.synchronize()
        self.lanes = []
        for p in self._raw:
            _lib.lib().tfc_stream_destroy(p)
        self._raw = []

    def __del__(self):
        try:
            for p in getattr(self, "_raw", []):
                _lib.lib().tfc_stream_destroy(p)
        except Exception:
            pass
