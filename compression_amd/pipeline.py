"""Stream plumbing for model pipelines on one GPU: several batches in flight.

A model step is transforms (SignalConv2D / GDN: every CU for tens of milliseconds) plus range coding (a few waves
for tens of milliseconds, bound by the length of one stream's chain whatever the batch).  One behind the other a
step costs their sum; with several steps in flight on ordinary HIP streams the hardware packs the kernels of
different steps onto the chip (CU-masked stream pairs and a software pipeline over one masked pair
were measured slower in round 3 and are gone).  The reference has no counterpart — its ops share TensorFlow's
inter/intra-op thread pools (range_coder_kernels.cc:212-215).

Nothing here synchronises with the host; a `Lane` only orders its streams with events.
"""
from __future__ import annotations

import contextlib
import ctypes as C

import torch

from . import _lib

__all__ = ["StepLanes", "Lane", "inline_lane", "chip_shared", "cached_bytes", "empty_cache"]


def cached_bytes() -> int:
    """Device memory the HIP library keeps for reuse (tfc_cache_bytes)."""
    n = C.c_longlong()
    _lib.check(_lib.lib().tfc_cache_bytes(C.byref(n)))
    return int(n.value)


def empty_cache() -> int:
    """Hands the library's idle cached blocks back to the driver (tfc_cache_trim) -> bytes released; the library-side
    counterpart of torch.cuda.empty_cache()."""
    n = C.c_longlong()
    _lib.check(_lib.lib().tfc_cache_trim(C.byref(n)))
    return int(n.value)


@contextlib.contextmanager
def chip_shared(shared=True):
    """Coder handles created inside know that other kernels run beside theirs (tfc_set_chip_shared): batches of 512
    streams and more are packed two waves per SIMD, on half the CUs."""
    before = _lib.lib().tfc_set_chip_shared(1 if shared else 0)     # -> the previous value (nested uses restore it)
    try:
        yield
    finally:
        _lib.lib().tfc_set_chip_shared(before)


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


class StepLanes:
    """`depth` lanes of one ordinary HIP stream each: whole steps overlap.  Create it once, early in the process:
    which hardware queue a stream lands on depends on what created streams before it (profiles/r03_notes.md), so a
    server makes its lanes first and keeps them."""

    def __init__(self, depth=6, device=None):
        _lib.require_device()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.lanes = []
        for _ in range(int(depth)):
            s = torch.cuda.Stream(device=self.device)
            self.lanes.append(Lane(s, s))

    def lane(self, k):
        return self.lanes[k % len(self.lanes)]

    def synchronize(self):
        for lane in self.lanes:
            lane.transform.synchronize()
