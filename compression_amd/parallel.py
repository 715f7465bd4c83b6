"""Multi-GPU plumbing: the hot path shards over the batch dimension (one code stream
per image, no cross-stream state — cc/kernels/range_coder_kernels.cc:225-226), so the
only communication is the batch split and a variable-length gather of the coded bytes.
One process per GPU; `torch.distributed` with backend "nccl" (= RCCL over xGMI) on
GPUs, "gloo" in the CPU tests."""
from __future__ import annotations

import torch
import torch.distributed as dist

__all__ = ["shard_range", "gather_encoded", "gather_encoded_async", "GatheredStrings", "broadcast_tables"]


def shard_range(total: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of `total` units for `rank` (sizes differ by at most 1)."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_encoded(blob: torch.Tensor, offsets: torch.Tensor, group=None):
    """All-gathers per-rank packed byte strings.

    blob: uint8 [total_bytes_r], offsets: int64 [streams_r + 1] (as produced by
    tfc_encoder_finalize).  Returns (blob_all uint8, offsets_all int64 [sum streams + 1])
    with ranks concatenated in rank order — identical to what a single process coding
    the whole batch would have produced.  Two collectives: lengths, then padded bytes."""
    world = dist.get_world_size(group)
    device = blob.device
    lengths = (offsets[1:] - offsets[:-1]).to(torch.int64)
    meta = torch.tensor([lengths.numel(), int(offsets[-1])], dtype=torch.int64, device=device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    max_streams = max(int(m[0]) for m in metas)
    max_bytes = max(int(m[1]) for m in metas)
    pad_len = torch.zeros(max_streams, dtype=torch.int64, device=device)
    pad_len[:lengths.numel()] = lengths
    pad_blob = torch.zeros(max(max_bytes, 1), dtype=torch.uint8, device=device)
    pad_blob[:blob.numel()] = blob
    all_len = [torch.zeros_like(pad_len) for _ in range(world)]
    all_blob = [torch.zeros_like(pad_blob) for _ in range(world)]
    dist.all_gather(all_len, pad_len, group=group)
    dist.all_gather(all_blob, pad_blob, group=group)
    lens = torch.cat([all_len[r][:int(metas[r][0])] for r in range(world)])
    blobs = torch.cat([all_blob[r][:int(metas[r][1])] for r in range(world)])
    offs = torch.zeros(lens.numel() + 1, dtype=torch.int64, device=device)
    offs[1:] = torch.cumsum(lens, 0)
    return blobs, offs


class GatheredStrings:
    """What gather_encoded_async leaves on every rank: the ranks' strings side by side, each rank's bytes in a slot of
    `capacity_bytes`, its per-stream lengths in a slot of `capacity_streams`.  Stream i of rank r is
    blob[r, starts[r, i] : starts[r, i] + lengths[r, i]]; `counts[r]` streams, `totals[r]` bytes.  `overflow` (device
    bool): some rank had more bytes or streams than its slot takes — then packed() gathers again, exactly.  Nothing in
    here has been looked at by the host."""

    def __init__(self, blob, lengths, counts, totals, overflow, exact, works=()):
        self.blob, self.lengths, self.counts, self.totals, self.overflow = blob, lengths, counts, totals, overflow
        self._exact = exact          # () -> (blob_all, offsets_all) by the synchronising path
        self._works = list(works)    # the collectives in flight (async_op): wait() orders the current stream behind them

    def wait(self):
        """Orders the CURRENT stream behind the two collectives (no host synchronisation with RCCL); the stream that
        enqueued the gather was never made to wait for them."""
        for w in self._works:
            w.wait()
        self._works = []

    @property
    def starts(self):
        return torch.cumsum(self.lengths, 1) - self.lengths

    def packed(self):
        """(blob_all uint8, offsets_all int64) as gather_encoded returns them — ranks concatenated in rank order,
        identical to one process coding the whole batch.  Synchronises (the host needs the sizes)."""
        self.wait()
        if bool(self.overflow):
            return self._exact()
        counts = [int(c) for c in self.counts.tolist()]
        totals = [int(t) for t in self.totals.tolist()]
        lens = torch.cat([self.lengths[r, :counts[r]] for r in range(len(counts))])
        blobs = torch.cat([self.blob[r, :totals[r]] for r in range(len(totals))])
        offs = torch.zeros(lens.numel() + 1, dtype=torch.int64, device=lens.device)
        offs[1:] = torch.cumsum(lens, 0)
        return blobs, offs


def gather_encoded_async(blob: torch.Tensor, offsets: torch.Tensor, capacity_bytes: int, capacity_streams: int = 0,
                         group=None) -> GatheredStrings:
    """The variable-length gather of SURVEY 8(e) without a host synchronisation: every rank contributes a slot of
    FIXED capacity (`capacity_bytes`, `capacity_streams` — host-side numbers: a model's streams per rank, and a byte
    bound the caller keeps from the steps it has already retired), so the two all-gathers (lengths + totals; padded
    bytes) are enqueued with sizes the host knows and nothing is read back: a step in flight ends with its gather
    and the host goes on enqueuing the next one.  Bytes beyond a slot are cut and flagged (GatheredStrings.overflow).
    The collectives are asynchronous also for the enqueuing STREAM (async_op): RCCL runs them on its own stream behind
    what the current stream holds so far, and the current stream is not made to wait for them — the steps in flight on
    other streams share that one RCCL stream, and a step whose next kernels waited for its gather would wait for every
    gather enqueued before it, i.e. for the other steps' encoders (measured on one GPU with a world of 1: bls2017 10.0
    instead of 5.9 ms per step); GatheredStrings.wait() / packed() order a consumer behind them.
    blob: uint8 [>= offsets[-1]] (a handle's device_strings view at slab capacity is fine), offsets int64 [streams + 1]."""
    world = dist.get_world_size(group)
    device = blob.device
    streams = offsets.numel() - 1
    cs = max(int(capacity_streams), streams, 1)
    cb = max(int(capacity_bytes), 1)
    # [lengths ... | streams | total bytes]: one collective for all the integers
    meta = torch.zeros(cs + 2, dtype=torch.int64, device=device)
    meta[:streams] = offsets[1:] - offsets[:-1]
    meta[cs] = streams
    meta[cs + 1] = offsets[-1]
    pad = torch.zeros(cb, dtype=torch.uint8, device=device)
    n = min(cb, blob.numel())
    pad[:n] = blob[:n]            # (bytes past offsets[-1] inside the slot travel too: the receiver cuts at the total)
    metas = torch.empty(world * (cs + 2), dtype=torch.int64, device=device)
    blobs = torch.empty(world * cb, dtype=torch.uint8, device=device)
    works = [dist.all_gather_into_tensor(metas, meta, group=group, async_op=True),      # (flat outputs: the form gloo takes as well)
             dist.all_gather_into_tensor(blobs, pad, group=group, async_op=True)]
    metas, blobs = metas.view(world, cs + 2), blobs.view(world, cb)
    return _LazyGathered(blobs, metas, cs, cb, lambda: gather_encoded(blob[:int(offsets[-1])], offsets, group=group), works)


class _LazyGathered(GatheredStrings):
    """(the fields derived from the gathered integers are formed behind wait(): forming them at once would make the
    enqueuing stream read what the collective has not written yet)"""

    def __init__(self, blobs, metas, cs, cb, exact, works):
        super().__init__(blobs, None, None, None, None, exact, works)
        self._metas, self._cs, self._cb = metas, cs, cb

    def wait(self):
        super().wait()
        if self.lengths is None:
            m, cs = self._metas, self._cs
            self.lengths, self.counts, self.totals = m[:, :cs], m[:, cs], m[:, cs + 1]
            self.overflow = (self.totals > self._cb).any() | (self.counts > cs).any()


def broadcast_tables(module: torch.nn.Module, src: int = 0, group=None):
    """Replicates weights and range-coding tables from `src` (tables must be shared, never
    rebuilt per rank: continuous_base.py:175-184)."""
    backend = dist.get_backend(group)
    for t in list(module.parameters()) + list(module.buffers()):
        if backend == "nccl" and not t.is_cuda:
            # the range-coding tables live on the host (the coder library keeps its own device copy):
            # through the device for RCCL, back into the same storage
            staged = t.data.to(torch.device("cuda", torch.cuda.current_device()))
            dist.broadcast(staged, src=src, group=group)
            t.data.copy_(staged.cpu())
        else:
            dist.broadcast(t.data, src=src, group=group)
    from .ops import gen_ops
    gen_ops.invalidate_table_cache()        # `.data` writes do not advance the version counters the caches key on
    for m in module.modules():
        if hasattr(m, "invalidate_kernel_cache"):
            m.invalidate_kernel_cache()
        for name in ("_dev_cache", "_dev_offsets"):
            if name in m.__dict__:
                object.__setattr__(m, name, None)
