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
import time

import torch

from . import _lib

__all__ = ["CoderPartition", "Lane", "inline_lane", "SoftwarePipeline", "chip_shared", "cached_bytes", "empty_cache"]


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
    _lib.check(_lib.lib().tfc_set_chip_shared(1 if shared else 0))
    try:
        yield
    finally:
        _lib.lib().tfc_set_chip_shared(0)


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


class SoftwarePipeline:
    """Steps whose stages alternate between transform work and coding, software-pipelined over ONE transform
    stream and ONE coder stream (a `CoderPartition(depth=1)` lane: disjoint CU sets).

    A step is a list of (kind, fn) stages, kind "transform" or "coder" (models' `codec_stages`): `head`
    transform stages, then coder stages with transform stages in between, then `tail` transform stages.  The
    coder stream runs the coding stages of consecutive steps back to back.  The transform stream runs, beside
    step k's LAST coding stage, the tail of step k - 1 and the head of step k + 1; if a step has two coding
    stages and a tail of several stages, the first tail stage of step k - 1 runs beside step k's FIRST coding stage
    instead (it has to be shorter than that stage: the transform stage between the two coding stages queues up
    behind it).  Each stream's FIFO order is the schedule; nothing here synchronises with the host.

    Transform stages that run beside a coding stage are released by the library's coder gate
    (tfc_set_coder_gate: an event recorded immediately in front of the stage's first long coding kernel), i.e.
    together with the coding kernel they run beside, never ahead of it: a coding kernel that becomes ready while
    the transform queue is running large grids back to back is not dispatched until that queue drains
    (tools/queue_pair_probe.py, profiles/r03_notes.md).
    """

    def __init__(self, lane: Lane):
        self.lane = lane
        self.T, self.C = lane.transform, lane.coder
        self._deferred = None                # (tail stages of the previous step, event behind its last coder stage)
        self._last_gate = None               # gate of the previous step's last coding stage
        self.host_log = None                 # a list: (stage name, host time entering, leaving) per enqueued stage

    def _run(self, stream, waits, fn, gate=None):
        for ev in waits:
            if ev is not None:
                stream.wait_event(ev)
        t0 = time.perf_counter() if self.host_log is not None else 0.0
        with torch.cuda.stream(stream):
            if gate is not None:
                gate.record(stream)          # creates the HIP event; re-recorded by the library further down the stream
                _lib.check(_lib.lib().tfc_set_coder_gate(gate.cuda_event))
            try:
                out = fn()
            finally:
                if gate is not None:
                    _lib.lib().tfc_set_coder_gate(None)
        ev = torch.cuda.Event()
        ev.record(stream)
        if self.host_log is not None:
            self.host_log.append((getattr(fn, "__name__", "?"), t0, time.perf_counter()))
        return out, ev

    def _tail(self, stages, dep, gate):
        """Runs deferred tail stages on the transform stream: behind their own step's coding (`dep`), beside the
        coding kernel `gate` stands in front of."""
        out = ev = None
        for _, fn in stages:
            out, ev = self._run(self.T, [dep, gate], fn)
            dep = gate = None
        return out, ev

    def submit(self, stages, after=None):
        """Enqueues one step up to and including its last coding stage, and the tail of the previous step.
        Returns (what the previous step's last stage returned, its end event) — (None, None) for the first."""
        stages = list(stages)
        kinds = [k for k, _ in stages]
        assert kinds[0] == "transform" and kinds[-1] == "transform" and "coder" in kinds
        first_c = kinds.index("coder")
        last_c = len(kinds) - 1 - kinds[::-1].index("coder")
        start = None
        if after is not None:
            start = torch.cuda.Event()
            start.record(after)
        ev = None
        for _, fn in stages[:first_c]:
            # head: released with the previous step's last coding kernel, not ahead of it
            _, ev = self._run(self.T, [start, self._last_gate], fn)
            start = None
        prev, done = self._deferred, (None, None)
        split = prev is not None and last_c > first_c and len(prev[0]) > 1
        for i in range(first_c, last_c + 1):
            kind, fn = stages[i]
            if kind == "coder" and i in (first_c, last_c):
                gate = torch.cuda.Event()
                _, ev = self._run(self.C, [ev], fn, gate=gate)
                if i == first_c and split:
                    self._tail(prev[0][:1], prev[1], gate)
                    prev = (prev[0][1:], None)
                if i == last_c:
                    if prev is not None:
                        done = self._tail(prev[0], prev[1], gate)
                    self._last_gate = gate
            else:
                _, ev = self._run(self.C if kind == "coder" else self.T, [ev], fn)
        self._deferred = (stages[last_c + 1:], ev)
        return done

    def drain(self):
        """Enqueues the last step's tail; returns (its result, its end event)."""
        if self._deferred is None:
            return None, None
        prev, self._deferred = self._deferred, None
        return self._tail(prev[0], prev[1], None)
