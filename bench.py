#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): Mpixels/s of the entropy-coding round trip
(range encode + range decode, bit-exact) on MI355X, at BASELINE config 2:
512 code streams x (16*16*192 = 49152) int32 latents = 512 images of 256x256,
192 discretised-Gaussian tables of precision 12 (escape coding enabled).

One "step" = one full pass of the hot path over the batch, inputs resident in
HBM: CreateRangeEncoder -> EntropyEncodeChannel -> EntropyEncodeFinalize ->
CreateRangeDecoder -> EntropyDecodeChannel -> EntropyDecodeFinalize, all through
the C ABI (libtfc_hip.so).  Prints ONE JSON line on rank 0.

Steps are independent batches, so they go to the GPU in groups of `--inflight` (default 20, the headline's FIXED
operating point: more steps are more groups of 20, not larger groups) as one launch per stage, all enqueued by ONE
host thread through the library's stream-ordered path (throughput-mode handles: one code stream per lane, range
errors deferred, finalize on the device, the decoder reads the encoder's device-resident strings): a 512-stream
step is 8 waves, so the chip only fills with many steps resident at once.  Every slot of a group codes its own seeded
tensor.  The timed region contains EXACTLY `--steps` complete steps; `single_batch` in the output is one step at a
time (BASELINE config 2 as literally written), `saturation` the rate at 1 ... 128 batches per group.  After the timed
region every step's decoded tensor is compared with its input, and the bytes of EVERY distinct input with the CPU
reference's bytes (sha256 over all 512 streams of each).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N ...            (no launcher around it: starts the N ranks itself, fails without N devices)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: the batch dimension shards (independent code streams), so every rank
codes its own 512-stream shard (weak scaling); no data-path collective, only the
barrier + max-reduce of the timing.

The same line carries, as sub-objects (N = 1 only; `--no-extras` drops them):
  single_batch  ONE 512-stream batch at a time (BASELINE config 2 as literally written), both families;
  escapes       the same --steps command on two other input laws: 1 % of the symbols replaced by far-out values
                (SURVEY.md §8d's second run) and the overflow draws folded away (no escape codes at all);
                slot 0's bytes compared with the CPU reference (the headline itself is on §8d's law:
                draws inverted through the tables, overflow bucket included);
  models        BASELINE config 1 (bls2017, 512 x 256x256) and config 4 (bmshj2018, 128 x 768x512) full
                compress + decompress, several steps in flight on ordinary streams (compression_amd/pipeline.py),
                every image's strings compared with the CPU reference coder on the model's own symbols;
  conv          SignalConv2D TFLOP/s per layer shape of config 4;
  saturation    batches per launch group 1, 2, 4, 8, 20, 64, 128 -> Mpixels/s, kernel times, algorithmic GB/s;
  gdn_fwd       BASELINE config 3, cold (rotating tensors): GDN / IGDN, forward / backward, (1, 1) and (2, 1/2).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and
# kernels that share a queue serialise: with the steps in flight below (plus torch's own streams)
# that capped the round trip at ~10 Gpixels/s; 16 queues lift it to ~12.5-13.  Read at runtime
# initialisation, so it has to be in the environment before torch touches the device.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import compression_amd as tfc  # noqa: E402
from compression_amd import _lib, synthetic  # noqa: E402

STREAMS = 512
CHANNELS = 192
ELEMS = 16 * 16 * CHANNELS           # latents of one 256x256 image (bls2017 geometry)
PIXELS_PER_STREAM = 256 * 256
PRECISION = 12
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: 8 TB/s spec


def build_tables(device, num_tables=CHANNELS, octave=24.0):
    """192 Gaussian tables through the PRODUCT table builder (HIP pmf_to_quantized_cdf)."""
    pmfs, _ = synthetic.gaussian_pmfs(num_tables=num_tables, octave=octave)
    # PmfToQuantizedCdf takes a rectangular [rows, n] tensor: one call per distinct row length
    by_len = {}
    for i, p in enumerate(pmfs):
        by_len.setdefault(len(p), []).append(i)
    cdfs = [None] * len(pmfs)
    for n, rows in by_len.items():
        c = tfc.pmf_to_quantized_cdf(torch.from_numpy(np.stack([pmfs[i] for i in rows])).to(device), PRECISION)
        for i, row in zip(rows, c.cpu().numpy()):
            cdfs[i] = row
    return synthetic.assemble_lookup(cdfs, PRECISION, overflow=True)


def profile_query(name):
    ms = C.c_double()
    n = C.c_int64()
    _lib.lib().tfc_profile_query(name.encode(), C.byref(ms), C.byref(n))
    return ms.value, n.value


def one_step(lookup_t, value_t, mode):
    """CreateRangeEncoder .. EntropyDecodeFinalize on the current HIP stream, nothing read back."""
    return step_group(lookup_t, [value_t], mode)[0]


def step_group(lookup_t, values, mode, host=None):
    """len(values) independent steps on the current HIP stream, nothing read back.  Every step has its own
    handles and strings; the coding calls of the group go to the GPU as one launch per stage
    (entropy_encode_channel_many / entropy_decode_channel_many): the hardware overlaps only ~8 kernels
    however many streams carry them, and one 512-stream call is 8 waves.  Creation and finalisation of the
    handles are batched the same way (one allocation and one small launch per group instead of per handle).
    `host` (a HostSink): the strings are also copied to pinned host memory, on a second stream, while the
    decoders run (the reference's ops return host strings)."""
    hs = tfc.create_range_encoders(len(values), [STREAMS], lookup_t, mode=mode, deferred_errors=True)
    hs = tfc.entropy_encode_channel_many(hs, values)
    hs = tfc.entropy_encode_finalize_device_many(hs)
    encoded = torch.cuda.Event()
    encoded.record()
    ds = tfc.create_range_decoders(hs, lookup_t, mode=mode)
    ds, decoded = tfc.entropy_decode_channel_many(ds, [ELEMS], torch.int32)
    oks = tfc.entropy_decode_finalize_device_many(ds)
    if host is not None:
        host.fetch(hs, encoded)
    return list(zip(hs, ds, decoded, oks))


class HostSink:
    """Pinned host buffers + a copy stream: the strings of a group of finalized encoder handles are copied out of
    HBM asynchronously — offsets first (the host needs the byte counts), then exactly the bytes."""

    def __init__(self, device, slots):
        self.stream = torch.cuda.Stream(device=device)
        self.offsets = [torch.empty(STREAMS + 1, dtype=torch.int64).pin_memory() for _ in range(slots)]
        self.blobs = [torch.empty(3 * STREAMS * ELEMS // 2, dtype=torch.uint8).pin_memory() for _ in range(slots)]
        self.bytes = 0

    def fetch(self, handles, encoded):
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(encoded)
            views = [tfc.device_strings(h) for h in handles]
            for k, (_, off) in enumerate(views):
                self.offsets[k].copy_(off, non_blocking=True)
            self.stream.synchronize()          # the host waits for the ENCODER only; the decoders are already enqueued
            for k, (blob, _) in enumerate(views):
                n = int(self.offsets[k][-1])
                self.blobs[k][:n].copy_(blob[:n], non_blocking=True)
                self.bytes += n


def sample_symbols_device(lookup, seed, device, escape_fraction=0.0, fold=False):
    """[STREAMS, ELEMS] int32 symbols on the device, the input law of SURVEY.md 8(d): uniform `precision`-bit draws
    inverted through each channel's CDF (same law as synthetic.sample_symbols, drawn with torch's generator so that
    32 slots take milliseconds instead of minutes).  A draw that lands in a row's overflow bucket IS the row's escape
    symbol — a value just outside the table, coded with the shortest Elias-gamma code — so the tables' own tail mass
    (2^-8 per row, ~0.4 % of the symbols) takes the escape path.  `escape_fraction` > 0 additionally replaces that
    share of the symbols by +-(len + Geometric(0.2)) (8(d)'s second run); `fold=True` moves the overflow draws onto
    the neighbouring plain symbol instead (an escape-free input, for comparison only)."""
    rows = synthetic.lookup_rows(lookup)
    ntab = len(rows)
    width = max(len(c) for _, c in rows)
    table = np.full((ntab, width), np.iinfo(np.int32).max, np.int64)
    for t, (_, c) in enumerate(rows):
        table[t, :len(c)] = c
    nsym = torch.tensor([len(c) - 1 for _, c in rows], device=device)
    esc = torch.tensor([sp < 0 for sp, _ in rows], device=device)
    table_t = torch.from_numpy(table).to(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + seed)
    u = torch.randint(0, 1 << PRECISION, (STREAMS, ELEMS), generator=gen, device=device)
    col_tab = torch.arange(ELEMS, device=device) % ntab
    # searchsorted per channel: bring the channel axis first
    reps = ELEMS // ntab
    uu = u.view(STREAMS, reps, ntab).permute(2, 0, 1).reshape(ntab, -1).contiguous()
    sym = torch.searchsorted(table_t, uu, right=True) - 1
    top = (torch.where(esc, nsym - 2, nsym - 1) if fold else nsym - 1).clamp(min=0)[:, None]
    sym = torch.minimum(sym, top).clamp(min=0)
    out = sym.view(ntab, STREAMS, reps).permute(1, 2, 0).reshape(STREAMS, ELEMS)
    if escape_fraction > 0:
        mask = torch.rand((STREAMS, ELEMS), generator=gen, device=device) < escape_fraction
        mask &= esc[col_tab][None, :]
        lens = (nsym - 1)[col_tab][None, :]
        geo = torch.empty((STREAMS, ELEMS), device=device).geometric_(0.2, generator=gen).long()
        neg = torch.randint(0, 2, (STREAMS, ELEMS), generator=gen, device=device).bool()
        out = torch.where(mask, torch.where(neg, -geo, lens + geo), out)
    return out.to(torch.int32).contiguous()


def escape_share(lookup, value_t):
    """Share of the symbols of `value_t` [STREAMS, ELEMS] that take the escape path of their row."""
    rows = synthetic.lookup_rows(lookup)
    limit = torch.tensor([len(c) - 2 if sp < 0 else 1 << 30 for sp, c in rows], device=value_t.device)
    lim = limit[torch.arange(value_t.shape[1], device=value_t.device) % len(rows)][None, :]
    return float(((value_t < 0) | (value_t >= lim)).float().mean().item())


GDN_ROTATE = 4      # input / output pairs a timed loop rotates over: 4 x (101 + 101 MB) = 805 MB, the Infinity Cache is 256 MiB


def gdn_forward_bandwidth(device, steps=20):
    """BASELINE config 3 (second half of the metric): GDN and IGDN, forward and backward, (alpha, epsilon) = (1, 1) and
    (2, 1/2), on 256 x 192 x 32 x 32 bf16 (NHWC [262144, 192]).  Algorithmic bytes: forward read x + write y; backward
    read x, g + write dx.  COLD numbers: every timed loop rotates over GDN_ROTATE distinct input tensors and as many
    distinct outputs (kept alive, so the allocator hands out different blocks), 805 MB per rotation against 256 MiB of
    Infinity Cache — no launch can find its input (or the lines it writes) in a cache; `warm` is the old measurement
    (one input, 20 launches back to back: its 201 MB working set fits the Infinity Cache) for comparison."""
    from compression_amd.layers import functional, gdn_backward, gdn_forward
    torch.manual_seed(3)
    C, M = 192, 256 * 32 * 32
    xs = [torch.randn(M, C, device=device).bfloat16() for _ in range(GDN_ROTATE)]
    gs = [torch.randn(M, C, device=device).bfloat16() for _ in range(GDN_ROTATE)]
    beta = (1 + 0.1 * torch.rand(C)).to(device)
    gamma = (0.1 * torch.eye(C) + 0.01 * torch.rand(C, C)).to(device)
    # inference form: the kernels' image of (beta, gamma) is prepared once (tfc_gdn_params_create), every call is
    # one launch of the forward kernel
    prepared = functional.GDNPrepared(beta, gamma, torch.bfloat16)
    nbytes = 2 * xs[0].numel() * xs[0].element_size()
    bwd_bytes = 3 * xs[0].numel() * xs[0].element_size()

    def fwd_ms(inverse, alpha, epsilon, rotate):
        ys = [None] * GDN_ROTATE
        call = lambda k: gdn_forward(xs[k % rotate], beta, gamma, inverse=inverse, alpha=alpha, epsilon=epsilon, prepared=prepared)
        for k in range(GDN_ROTATE):
            ys[k] = call(k)
        torch.cuda.synchronize()
        # HIP events over the timed region, on the launch stream: `steps` launches back to back.  Twice, the faster
        # counts: replacing an output while the old one is still referenced makes torch's allocator fetch one more
        # 100 MB block the first time round (a hipMalloc of ~20 ms inside the loop: 1.04 ms "per launch" in one line).
        best = None
        for _ in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for k in range(steps):
                ys[k % rotate] = call(k)
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1) / steps
            best = ms if best is None else min(best, ms)
        return best

    def bwd_ms(inverse, alpha, epsilon):
        keep = [None] * GDN_ROTATE
        for k in range(GDN_ROTATE):
            keep[k] = gdn_backward(xs[k], gs[k], beta, gamma, inverse=inverse, alpha=alpha, epsilon=epsilon)
        torch.cuda.synchronize()
        _lib.lib().tfc_profile_enable(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(steps):
            keep[k % GDN_ROTATE] = gdn_backward(xs[k % GDN_ROTATE], gs[k % GDN_ROTATE], beta, gamma,
                                                inverse=inverse, alpha=alpha, epsilon=epsilon)
        e1.record()
        e1.synchronize()
        passes = {}
        for name in ("gdn_backward_fused", "gdn_backward_t", "gdn_backward_dx", "gdn_backward_params"):
            pms, pn = profile_query(name)
            if pn:
                passes[name] = round(pms / pn, 4)
        _lib.lib().tfc_profile_enable(0)
        # (the library's per-launch event pairs serialise the launches; their sum is the kernels' own time)
        return sum(passes.values()), passes, e0.elapsed_time(e1) / steps

    def copy_ms(rotate):
        """A plain device copy of the same tensors (torch's elementwise copy kernel: read x, write y — GDN's
        algorithmic traffic with no arithmetic), timed the same way: the HBM rate this box gives a 2-tensor stream."""
        ys = [torch.empty_like(xs[0]) for _ in range(GDN_ROTATE)]
        for k in range(GDN_ROTATE):
            ys[k].copy_(xs[k % rotate])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(steps):
            ys[k % rotate].copy_(xs[k % rotate])
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / steps

    fwd_ms(False, 1, 1, GDN_ROTATE)           # (clocks and allocator warm before the first figure)
    copy_cold, copy_warm = copy_ms(GDN_ROTATE), copy_ms(1)
    variants = {}
    for name, inverse, alpha, epsilon in (("gdn", False, 1, 1), ("igdn", True, 1, 1),
                                          ("gdn_alpha2_eps0.5", False, 2, 0.5), ("igdn_alpha2_eps0.5", True, 2, 0.5)):
        cold = fwd_ms(inverse, alpha, epsilon, GDN_ROTATE)
        warm = fwd_ms(inverse, alpha, epsilon, 1)
        kms, passes, loop_ms = bwd_ms(inverse, alpha, epsilon)
        variants[name] = {
            "forward": {"kernel_ms": round(cold, 4), "achieved": round(nbytes / 1e6 / cold, 1),
                        "frac": round(nbytes / 1e6 / cold / HBM_PEAK_GBS, 4),
                        "warm_kernel_ms": round(warm, 4), "warm_achieved": round(nbytes / 1e6 / warm, 1)},
            "backward": {"kernel_ms": round(kms, 4), "passes_ms": passes, "loop_ms_per_call": round(loop_ms, 4),
                         "achieved": round(bwd_bytes / 1e6 / kms, 1) if kms else None,
                         "frac": round(bwd_bytes / 1e6 / kms / HBM_PEAK_GBS, 4) if kms else None}}
    # (b) the library's own timers for the headline variant: an event pair around every single launch (includes the launch latency)
    _lib.lib().tfc_profile_enable(1)
    ys = [None] * GDN_ROTATE
    for k in range(steps):
        ys[k % GDN_ROTATE] = gdn_forward(xs[k % GDN_ROTATE], beta, gamma, prepared=prepared)
    torch.cuda.synchronize()
    ms, n = profile_query("gdn_forward")
    _lib.lib().tfc_profile_enable(0)
    head = variants["gdn"]
    return {"workload": "GDN fwd, [262144, 192] bf16 (= 256x192x32x32), alpha=1, eps=1, COLD: rotating over "
                        f"{GDN_ROTATE} input and {GDN_ROTATE} output tensors (805 MB per rotation > 256 MiB Infinity Cache)",
            "kernel_ms": head["forward"]["kernel_ms"], "kernel_ms_single_launch_events": round(ms / max(n, 1), 4),
            "timing": f"HIP events around {steps} back-to-back launches on the launch stream (parameters prepared once, "
                      "as the layer does under no_grad); kernel_ms_single_launch_events = an event pair around every launch",
            "algorithmic_bytes": nbytes,
            "achieved": head["forward"]["achieved"], "unit": "GB/s", "peak": HBM_PEAK_GBS,
            "frac": head["forward"]["frac"], "bound": "hbm",
            "warm": {"kernel_ms": head["forward"]["warm_kernel_ms"], "achieved": head["forward"]["warm_achieved"],
                     "frac": round(head["forward"]["warm_achieved"] / HBM_PEAK_GBS, 4),
                     "note": "one input tensor, launches back to back (rounds 1-4 measured this): the 201 MB working set "
                             "fits the 256 MiB Infinity Cache"},
            "copy_reference": {"cold_ms": round(copy_cold, 4), "cold_gbs": round(nbytes / 1e6 / copy_cold, 1),
                               "warm_ms": round(copy_warm, 4), "warm_gbs": round(nbytes / 1e6 / copy_warm, 1),
                               "frac_of_cold_copy": round(copy_cold / head["forward"]["kernel_ms"], 4),
                               "note": "torch's device copy of the same tensors, same rotation and timing: what this box "
                                       "streams for read x + write y with no arithmetic; frac_of_cold_copy = copy time / GDN time"},
            "traffic": pmc_traffic("gdn_fwd_bf16_kernel", GDN_PMC_PROFILE, GDN_SOURCES),
            "traffic_source": f"stored: profiles/{GDN_PMC_PROFILE} (rocprofv3 --pmc passes of this command on these sources; "
                              "null = taken on other sources)",
            "backward": dict(head["backward"], algorithmic_bytes=bwd_bytes, unit="GB/s"),
            "variants": {"note": "all cold (rotating tensors); forward algorithmic bytes = x + y, backward = x + g + dx; "
                                 "achieved in GB/s, frac of 8 TB/s; backward kernel_ms = sum of its kernels' own times",
                         **variants}}


CODER_SOURCES = ["compression_amd/csrc/range_coder.hip", "compression_amd/csrc/range_lanes.h", "compression_amd/csrc/range_pipe.h",
                 "compression_amd/csrc/range_encoder_fast.h", "compression_amd/csrc/range_decoder_fast.h"]
GDN_SOURCES = ["compression_amd/csrc/gdn.hip", "compression_amd/csrc/gdn_common.h",
               "compression_amd/csrc/gdn_backward.hip"]


def source_hashes(paths):
    """git blob hashes (sha1 of "blob <len>\\0" + content) of the kernel sources a profile was taken on."""
    import hashlib
    out = {}
    for rel in paths:
        try:
            data = open(os.path.join(ROOT, rel), "rb").read()
        except OSError:
            out[rel] = None
            continue
        out[rel] = hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()
    return out


def load_profile(name, sources, steps_per_launch=None):
    """A committed rocprofv3 --pmc summary (tools/pmc_summary.py / sq_summary.py stamp it with the git
    blob hashes of the kernel sources it was measured on), or None when the sources have changed since:
    a counter figure of other code is not a measurement of this run."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", name)))
    except OSError:
        return None
    stamp = table.get("_sources")
    now = source_hashes(sources)
    if not stamp or any(stamp.get(k) != v for k, v in now.items()):
        return None
    if steps_per_launch is not None and stamp.get("_steps_per_launch") != steps_per_launch:
        return None            # per-launch counters of a command that launched another number of steps
    return table


def pmc_traffic(kernel_substring, profile, sources, steps_per_launch=None):
    """HBM bytes per launch of a kernel from the committed rocprofv3 --pmc passes
    (tools/pmc_on_box.sh; same bench command).  Counter unit is KiB.  Corrections as
    MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE counts 128-B read requests as 64 B
    (x2; calibrated in the same passes on torch's fp32->bf16 copy of a 201 MB tensor: reported
    98322 KiB, true 196608 KiB), WRITE_SIZE is exact on that kernel's 98304 KiB output."""
    table = load_profile(profile, sources, steps_per_launch)
    if table is None:
        return None
    for name, ctrs in table.items():
        if kernel_substring in name and isinstance(ctrs, dict) and "FETCH_SIZE" in ctrs and "WRITE_SIZE" in ctrs:
            return int((2.0 * ctrs["FETCH_SIZE"]["mean"] + ctrs["WRITE_SIZE"]["mean"]) * 1024)
    return None


PMC_PROFILE = "r06_pmc_traffic.json"
GDN_PMC_PROFILE = "r06_pmc_gdn.json"
SQ_PROFILE = "r06_sq_inflight.json"


def valu_issue_floor(ms_per_step, kernels, steps_per_launch):
    """Vector-issue floor of a step: SQ_ACTIVE_INST_VALU (quad-cycles, summed over waves; committed
    rocprofv3 --pmc pass of this command, tools/sq_on_box.sh) of the encoder + decoder launches of one
    step, spread over all SIMDs = the time a step needs if every SIMD issued vector instructions without
    a gap.  None when the profile was taken on other sources."""
    table = load_profile(SQ_PROFILE, CODER_SOURCES, steps_per_launch)
    if table is None:
        return None
    quads = {}
    for name, row in table.items():
        for key in kernels:
            if key in name and key not in quads and isinstance(row, dict) and row.get("SQ_ACTIVE_INST_VALU"):
                quads[key] = row["SQ_ACTIVE_INST_VALU"] / steps_per_launch      # the launch codes that many steps
    if len(quads) != len(kernels):
        return None
    simds, clock_hz = 256 * 4, 2.4e9
    floor_ms = 1e3 * 4.0 * sum(quads.values()) / simds / clock_hz
    return {"valu_quad_cycles_per_step": {k: int(v) for k, v in quads.items()},
            "simds": simds, "clock_ghz": 2.4, "floor_ms_per_step": round(floor_ms, 4),
            "frac": round(floor_ms / ms_per_step, 4),
            "source": f"profiles/{SQ_PROFILE} (SQ_ACTIVE_INST_VALU, rocprofv3 --pmc)"}


def usable_cores():
    """Host cores this process may actually use: the affinity mask, capped by the cgroup CPU
    quota (a pool larger than the quota only gets throttled by CFS bandwidth control)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(float(q) / float(period) + 0.5))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = max(1, int(q / period + 0.5))
        except (OSError, ValueError):
            pass
    return n, (min(n, quota) if quota else n), quota


def cpu_baseline(lookup, slots, slot_shas):
    """Reference coder core (oracle/_ref) or its restatement on the host cores,
    sharded over streams like the reference's ThreadPool::ParallelFor.
    This is the ONLY place bench.py touches oracle/."""
    from oracle import oracle
    lib = oracle.best()
    hw_threads, cores, quota = usable_cores()
    # keep this process's own OpenMP / torch worker threads from spinning next to the pool
    torch.set_num_threads(1)
    time.sleep(1.0)
    value = slots[0].cpu().numpy()
    pixels = value.shape[0] * PIXELS_PER_STREAM
    best = None
    # the usable core count and half of it (SMT siblings): report the faster
    for threads in sorted({cores, max(cores // 2, 1)}, reverse=True):
        enc, dec, total, ok = lib.bench_roundtrip(lookup, value, threads=threads, reps=16)
        assert ok, "CPU baseline round trip failed"
        sums = (enc + dec)[1:]
        rt = float(np.median(sums))
        if best is None or rt < best[0]:
            k = int(np.argsort(sums)[len(sums) // 2])      # the median repetition
            best = (rt, threads, float(enc[1:][k]), float(dec[1:][k]), total, float(sums.min()))
    rt, threads, enc_s, dec_s, total, rt_min = best
    e1, d1, _, _ = lib.bench_roundtrip(lookup, value[:8], threads=1, reps=3)
    one_thread = (8 * PIXELS_PER_STREAM / 1e6) / float(np.median((e1 + d1)[1:]))
    # the bytes themselves: all streams of EVERY distinct input of the timed region, compared by hash with what the GPU
    # produced for the same input
    compared, identical, total = compare_slots_with_cpu(lib, lookup, slots, slot_shas, threads)
    same = compared == identical
    return {
        "value": round(pixels / 1e6 / rt, 2), "unit": "Mpixels/s", "cores": threads,
        "kind": lib.kind,
        "best_repetition_mpixels_s": round(pixels / 1e6 / rt_min, 2),
        "sample": f"full batch ({value.shape[0]} streams x {value.shape[1]} symbols) x 16 repetitions "
                  f"(first discarded), streams sharded over a persistent pool of {threads} host "
                  f"threads ({hw_threads} hardware threads visible, cgroup CPU quota "
                  f"{quota if quota else 'none'}), median encode+decode time",
        "encode_ms": round(1e3 * enc_s, 3), "decode_ms": round(1e3 * dec_s, 3),
        "one_thread_mpixels_s": round(one_thread, 2),
        "bytes_identical_to_gpu": bool(same),
        "slots_compared": compared, "slots_identical": identical,
        "bytes_compared": f"sha256 of the packed blob and of its offsets, every one of the {compared} distinct inputs of the "
                          f"timed region ({int(total)} bytes, {compared} x {value.shape[0]} streams)",
    }


def model_conv_flops(workload, batch, hw):
    """2 M K N of every SignalConv2D of compress + decompress (synthesis counted on
    its output grid, without the multiplies on inserted zeros)."""
    h, w = hw
    c = 192
    def conv(pix, k, cin, cout):
        return 2.0 * pix * k * k * cin * cout
    if workload == "bls2017":
        per = (conv(h * w / 16, 9, 3, c) + conv(h * w / 64, 5, c, c) + conv(h * w / 256, 5, c, c))
        return batch * 2 * per                       # analysis + synthesis (mirror)
    per = (conv(h * w / 4, 5, 3, c) + conv(h * w / 16, 5, c, c) + conv(h * w / 64, 5, c, c) + conv(h * w / 256, 5, c, c))
    hyper = conv(h * w / 256, 3, c, c) + conv(h * w / 1024, 5, c, c) + conv(h * w / 4096, 5, c, c)
    # compress: analysis + hyper analysis + hyper synthesis (indexes); decompress: hyper synthesis + synthesis
    return batch * (2 * per + 3 * hyper)


def calibrate_hyperprior(model, x, index_mean=20.0, index_std=12.0, side_std=3.0, escape_target=2.0 ** -8):
    """Random-init weights leave bmshj2018's scale indexes all 0 (one 4-symbol table): not the workload
    the indexed coder is for.  Rescales three linear layers so that, on the bench images, the side latent has
    standard deviation `side_std`, the hyper-synthesis output (the scale index field) mean `index_mean` /
    standard deviation `index_std` before clipping to [0, 63] — a field that varies in space, covers the 64
    tables and puts most of its mass on the narrow ones, as a trained model's does — and the main latent is
    scaled so that the share of its symbols that fall outside their table (escape codes) is the tables' own
    design tail mass 2^-8 (continuous_base.py tail_mass), as it is for data the tables fit.  Returns the
    histogram of the indexes actually coded and the escape share reached."""
    em = model.entropy_model
    with torch.no_grad():
        xs = x[:8].to(model.compute_dtype)
        def scale_layer(layer, f):
            for name in ("kernel_real", "kernel_imag", "kernel_variable", "bias"):
                t = getattr(layer, name, None)
                if t is not None:
                    t.mul_(f)
        rows = synthetic.lookup_rows(em.cdf.cpu().numpy())
        width = torch.tensor([len(c) - 2 for _, c in rows], device=x.device)          # plain symbols per table
        offs = em.cdf_offset.to(x.device)

        def fields():
            y = model.analysis_transform(xs)
            z = model.hyper_analysis_transform(torch.abs(y))
            z_hat = model.side_entropy_model.quantize(z)
            out = model.hyper_synthesis_transform(z_hat)[:, :y.shape[1], :y.shape[2], :]
            return y, z, out

        def escape_share(y, out):
            flat = em._flatten_indexes(em._normalize_indexes(out)).long()
            sym = torch.round(y.float()).to(torch.int32) - offs[flat]
            return float(((sym < 0) | (sym >= width[flat])).float().mean()), flat

        last = model.hyper_synthesis_transform.layer_2
        scale_layer(model.analysis_transform.layer_3, 1.0 / float(fields()[0].float().std()))     # start: unit latent
        for _ in range(12):
            # the three targets interact (|y| feeds the hyper path): a few rounds of fixed-point iteration
            y, z, out = fields()
            scale_layer(model.hyper_analysis_transform.layer_2, side_std / float(z.float().std()))
            y, z, out = fields()
            g = index_std / float(out.float().std())
            last.kernel_variable.mul_(g)
            last.bias.mul_(g).add_(index_mean - float(out.float().mean()) * g)
            y, z, out = fields()
            share, flat = escape_share(y, out)
            if abs(share - escape_target) < 0.1 * escape_target:
                break
            # escapes grow monotonically with the latent's scale: a damped multiplicative step towards the target
            scale_layer(model.analysis_transform.layer_3, min(2.0, max(0.5, (escape_target / max(share, 1e-6)) ** 0.25)))
        y, z, out = fields()
        share, flat = escape_share(y, out)
        hist = torch.bincount(flat.reshape(-1), minlength=model.num_scales).cpu().numpy()
    return hist, share


def model_symbols(model, handle, b0, b1, em=None):
    """The int32 symbols (and table indexes) the model's coder read for images [b0, b1) of a step — from
    the very tensors the encode call was given (`handle.coder_inputs`), so that the CPU reference codes
    the same symbols, not a recomputation of the transforms."""
    y, flat = handle.coder_inputs
    y = y[b0:b1]
    n = y.shape[0]
    em = em if em is not None else model.entropy_model
    if flat is not None:
        flat = flat[b0:b1].reshape(n, -1)
        sym = torch.round(y.float()).to(torch.int32).reshape(n, -1) - em.cdf_offset.to(y.device)[flat.long()]
        return em.cdf.cpu().numpy(), sym.cpu().numpy(), flat.cpu().numpy()
    off = em.quantization_offset
    if off is not None:
        # continuous_batched.py:370-380 in the bottleneck's dtype, as the kernel's SymQuant load does
        y = (y - off.to(y.device, y.dtype))
    yq = torch.round(y.float()).to(torch.int32)
    sym = yq.reshape(n, -1) - em.cdf_offset.to(y.device).repeat(yq[0].numel() // em.cdf_offset.numel())
    return em.cdf.cpu().numpy(), sym.cpu().numpy(), None


def side_strings_check(model, handle, strings, chunk=32):
    """bmshj2018's SIDE string (the hyper latent z, channel mode on the deep-factorized prior's tables) of every image,
    byte-compared with the CPU reference coder on the symbols the model coded."""
    from oracle import oracle
    lib = oracle.best()
    _, cores, _ = usable_cores()
    batch = handle.coder_inputs[0].shape[0]
    mine = [bytes(s) for s in strings.reshape(-1)]
    differing = nbytes = symbols = 0
    for b0 in range(0, batch, chunk):
        lookup, sym, _ = model_symbols(model, handle, b0, min(b0 + chunk, batch), em=model.side_entropy_model)
        cpu_strings, _, _ = lib.encode(lookup, sym, threads=cores)
        dec, ok = lib.decode(lookup, cpu_strings, sym.shape[1], threads=cores)
        assert ok.all() and (dec == sym).all()
        differing += sum(a != b for a, b in zip(mine[b0:b0 + len(cpu_strings)], cpu_strings))
        nbytes += sum(len(c) for c in cpu_strings)
        symbols += sym.size
    return {"images_compared": batch, "images_differing": int(differing), "bytes": int(nbytes), "symbols": int(symbols),
            "bytes_identical_to_gpu": differing == 0}


def model_cpu_baseline(model, handle, strings, hw, chunk=32):
    """The reference's coder (oracle/_ref, or its restatement) on the host cores, on the SAME symbols the
    model coded — the main latent stream of EVERY image of the batch, byte-compared with the GPU's strings.
    The transforms have no CPU leg here (the reference's are TensorFlow/Eigen, not installable): the figure
    is Mpixels/s of the entropy-coding part only.  This and cpu_baseline() are the only places bench.py
    touches oracle/."""
    from oracle import oracle
    lib = oracle.best()
    _, cores, _ = usable_cores()
    batch = handle.coder_inputs[0].shape[0]
    enc_s = dec_s = 0.0
    differing = 0
    escapes = total = 0
    mine = [bytes(s) for s in strings.reshape(-1)]
    rows = None
    for b0 in range(0, batch, chunk):
        lookup, sym, flat = model_symbols(model, handle, b0, min(b0 + chunk, batch))
        if rows is None:
            rows = np.array([len(c) - 2 for _, c in synthetic.lookup_rows(lookup)])     # plain symbols per table
        t0 = time.perf_counter()
        cpu_strings, _, _ = lib.encode(lookup, sym, index=flat, threads=cores)
        t1 = time.perf_counter()
        dec, ok = lib.decode(lookup, cpu_strings, sym.shape[1], index=flat, threads=cores)
        t2 = time.perf_counter()
        assert ok.all() and (dec == sym).all()
        enc_s += t1 - t0
        dec_s += t2 - t1
        differing += sum(a != b for a, b in zip(mine[b0:b0 + len(cpu_strings)], cpu_strings))
        width = rows[flat] if flat is not None else rows[np.arange(sym.shape[1]) % len(rows)][None, :]
        escapes += int(((sym < 0) | (sym >= width)).sum())
        total += sym.size
    return {"value": round(batch * hw[0] * hw[1] / 1e6 / (enc_s + dec_s), 2), "unit": "Mpixels/s", "cores": cores,
            "kind": lib.kind,
            "sample": f"main latent stream of all {batch} images (the symbols the model coded), range encode + "
                      f"decode only, {chunk} images per call",
            "encode_ms": round(1e3 * enc_s, 2), "decode_ms": round(1e3 * dec_s, 2),
            "symbols": int(total), "escape_fraction": round(escapes / max(total, 1), 5),
            "bytes_identical_to_gpu": differing == 0, "images_compared": batch, "images_differing": int(differing)}


def make_model(workload, dtype, device, batch, rank=0, calibrate=True):
    torch.manual_seed(0)
    if workload == "bls2017":
        model = tfc.models.BLS2017Model(num_filters=192, compute_dtype=dtype)
        batch, hw = batch or 512, (256, 256)
    else:
        model = tfc.models.BMSHJ2018Model(num_filters=192, compute_dtype=dtype)
        batch, hw = batch or 128, (512, 768)
    model = model.to(device).init_compression()
    base = torch.from_numpy(synthetic.lowpass_images(8, hw[0], hw[1], seed=2 + rank)).to(device)
    x = base.repeat((batch + 7) // 8, 1, 1, 1)[:batch].contiguous()
    hist = None
    if workload == "bmshj2018" and calibrate:
        hist, _ = calibrate_hyperprior(model, x)
    return model, x, batch, hw, hist


class StepRecord:
    """What one model step in flight keeps alive until it is retired."""
    def __init__(self, out, x_hat, oks, end):
        self.out, self.x_hat, self.oks, self.end = out, x_hat, oks, end
        self.strings = None
        self.gathered = None


class StepGather:
    """The collective a multi-GPU model step ends with: the coded strings of every rank's images on every
    rank — per encoder handle one all-gather of lengths and totals and one of the bytes in fixed-capacity slots
    (parallel.gather_encoded_async), enqueued on the step's own stream behind its kernels with nothing read back, so the
    steps in flight stay in flight.  The slot size is a host-side number: 5/4 of the largest per-rank total (+ 64 KB) the first
    (untimed, synchronising) gather of each handle position saw, agreed between the ranks; a step that outgrows it is
    flagged on the device and counted when it is retired."""

    def __init__(self, device, group=None):
        from compression_amd import parallel
        self.parallel, self.device, self.group = parallel, device, group
        self.caps = {}
        self.gathers = self.steps = self.overflows = 0
        self.events = []

    def __call__(self, handles):
        import torch.distributed as dist
        from compression_amd.ops import gen_ops
        out = []
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i, h in enumerate(handles):
            blob, off = gen_ops.device_strings(h)
            if i not in self.caps:
                total = off[-1:].clone()
                dist.all_reduce(total, op=dist.ReduceOp.MAX, group=self.group)
                self.caps[i] = (5 * int(total) // 4 + 2 * 65536 - 1) // 65536 * 65536
                out.append(self.parallel.gather_encoded(blob[:int(off[-1])], off, group=self.group))
            elif os.environ.get("TFC_BENCH_GATHER") == "local":
                # (diagnostic: the gather's device work without the collectives — what of a step's cost is RCCL's ordering)
                pad = torch.zeros(self.caps[i], dtype=torch.uint8, device=blob.device)
                n = min(self.caps[i], blob.numel())
                pad[:n] = blob[:n]
                out.append((pad.clone(), off.clone()))
            else:
                out.append(self.parallel.gather_encoded_async(blob, off, self.caps[i], h.streams, group=self.group))
            self.gathers += 1
        e1.record()
        self.events.append((e0, e1))
        self.steps += 1
        return out

    def retire(self, gathered):
        for g in gathered or ():
            if isinstance(g, self.parallel.GatheredStrings):
                g.wait()
                if bool(g.overflow):
                    self.overflows += 1

    def reset(self):
        self.gathers = self.steps = self.overflows = 0
        self.events = []

    def collective_ms_per_step(self):
        """Mean time between the events around a step's gathers on the step's stream (after the timed region)."""
        if not self.events:
            return None
        return sum(a.elapsed_time(b) for a, b in self.events) / len(self.events)


def run_model_steps(model, x, steps, lanes, fetch=True, group=1, gather=None):
    """`steps` compress + decompress passes over `x`, step k on lanes[k % len(lanes)], all enqueued by this
    one thread with nothing read back inside a step; a lane's previous step is retired (host waits for its
    end event, fetches its strings and sanity flags) before the lane is reused — while the other lanes'
    steps keep the GPU busy.  group > 1: `group` steps (differently rolled copies of `x`) are enqueued as one
    unit through the model's compress_many / decompress_many — one coder launch per direction for all of
    them.  `gather` (a StepGather, multi-GPU runs): every unit ends with the variable-length gather of its strings across
    the ranks, enqueued on the unit's stream.  Returns (seconds, last record)."""
    main = torch.cuda.current_stream()
    pending = [None] * len(lanes)
    last = None
    assert steps % group == 0
    units = steps // group
    xs = [x] if group == 1 else [torch.roll(x, k, 0) for k in range(group)]

    def retire(rec):
        if rec.end is not None:
            rec.end.synchronize()
        if fetch:
            rec.strings = [tfc.fetch_strings(h) for h in rec.out if isinstance(h, tfc.gen_ops.EncoderHandle)]
            for other in getattr(rec.x_hat, "_tfc_group", (None, ()))[1]:
                for h in other:
                    if isinstance(h, tfc.gen_ops.EncoderHandle):
                        tfc.fetch_strings(h)
            for ok in rec.oks:
                assert bool(ok.cpu().all()), "EntropyDecodeFinalize reported a failed stream"
        if gather is not None:
            gather.retire(rec.gathered)
        return rec


    t0 = time.perf_counter()
    for k in range(units):
        slot = k % len(lanes)
        if pending[slot] is not None:
            last = retire(pending[slot])
        lane = lanes[slot].begin(main)
        gathered = None
        enc_handles = lambda outs: [h for out in outs for h in out if isinstance(h, tfc.gen_ops.EncoderHandle)]
        if group == 1:
            out = model.compress(x, device_result=True, lane=lane)
            if gather is not None:
                # the strings exist: their gather goes out while the decoders run (the reference's compress() returns
                # its strings; decompress() is another call)
                with lane.on("coder"):
                    gathered = gather(enc_handles([out]))
            x_hat, oks = model.decompress(*out, defer_sanity=True, lane=lane)
        else:
            with lane.on("transform"):
                packed = model.compress_many(xs)
                if gather is not None:
                    gathered = gather(enc_handles(packed))
                x_hats, ok = model.decompress_many(packed)
            # the record of the unit: its first batch's result (the parity checks look at rec.out), the handles of
            # the other batches behind it (their strings are fetched with the record too)
            out = tuple(packed[0])
            x_hat, oks = x_hats[0], (ok if isinstance(ok, (list, tuple)) else [ok])
            x_hat._tfc_group = (x_hats, packed[1:])
        pending[slot] = StepRecord(out, x_hat, oks, lane.end_event())
        pending[slot].gathered = gathered
    order = [(k % len(lanes)) for k in range(max(0, units - len(lanes)), units)]
    for slot in order:
        if pending[slot] is not None:
            last = retire(pending[slot])
            pending[slot] = None
    torch.cuda.synchronize()
    return time.perf_counter() - t0, last


def model_bench(workload, dtype_name, device, batch=0, steps=6, warmup=2, lanes=None,
                cpu=True, rank=0, world=1, distributed=False, group=1, queue=2, inflight_warmup=0, passes=3):
    """Full compress + decompress of a target model on synthetic images (BASELINE configs 1/4/5).  `lanes`: the
    pipeline.StepLanes the steps are spread over (made first thing in the process: which hardware queue a stream gets
    depends on what created streams before it, profiles/r03_notes.md); `queue` steps enqueued per lane; `group`
    batches per coder launch (the models' compress_many / decompress_many)."""
    import torch.distributed as dist
    from compression_amd import parallel, pipeline
    from compression_amd.ops import gen_ops
    dtype = torch.bfloat16 if dtype_name == "bf16" else torch.float32
    step_lanes = lanes if lanes is not None else pipeline.StepLanes(6, device)
    # what the legs before this one left in the two block caches (torch's and the library's) goes back to the driver:
    # the float32 / 8-batch C4 passes need the room as fresh blocks of their own sizes
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    pipeline.empty_cache()
    torch.cuda.reset_peak_memory_stats(device)
    model, x, batch, hw, hist = make_model(workload, dtype, device, batch, rank)
    if distributed:
        parallel.broadcast_tables(model)
    side = torch.cuda.Stream(device=device)          # never the null stream: it would serialise the masked streams
    with torch.cuda.stream(side):
        # (a) a lone step at a time on the whole chip, per-kernel split (nothing else resident)
        inline = [pipeline.inline_lane()]
        run_model_steps(model, x, max(warmup, 1), inline)
        _lib.lib().tfc_profile_enable(1)
        lone_steps = 2
        lone_s, rec = run_model_steps(model, x, lone_steps, inline)
        kern = {name: profile_query(name) for name in ("enc_kernel", "dec_kernel", "conv2d", "gdn_forward")}
        _lib.lib().tfc_profile_enable(0)
        # the lone steps' blocks (this side stream's pool: no in-flight lane can reuse them) go back before the lanes fill theirs
        del rec
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        # (b) the timed region: `queue` steps enqueued per lane — a stream whose step has finished still has work while
        # the host retires that step (waits for its end event, fetches strings and flags) and enqueues the next
        lanes = list(step_lanes.lanes) * max(1, queue)
        group = group if hasattr(model, "compress_many") else 1
        # multi-GPU: every step ENDS with the variable-length gather of its strings, inside the timed region
        step_gather = StepGather(device) if distributed else None
        run = lambda n: run_model_steps(model, x, n, lanes, group=group, gather=step_gather)
        steps = max(group, steps - steps % group)
        _lib.lib().tfc_set_chip_shared(1)                   # several steps in flight: pipeline.chip_shared()
        # untimed: every lane once (its buffers and streams primed); `inflight_warmup` units for the slow float32 C4
        run((inflight_warmup or max(warmup, len(lanes), 2)) * group)
        torch.cuda.synchronize()
        if distributed:
            step_gather.reset()
            dist.barrier()
            torch.cuda.synchronize()
        # `passes` timed passes, each bracketed like the first; the median counts (a pass whose buffers the allocators
        # had to fetch first — seen on fresh boxes with the float32 C4 — is an outlier, not the figure)
        pass_s = []
        rec = None
        for k in range(max(1, passes)):
            if k:
                rec = None
                torch.cuda.synchronize()
                if distributed:
                    step_gather.reset()
                    dist.barrier()
                    torch.cuda.synchronize()
            e, rec = run(steps)
            pass_s.append(e)
        elapsed = sorted(pass_s)[len(pass_s) // 2]
        _lib.lib().tfc_set_chip_shared(0)
        gathered = None
        collective = None
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()
            gathered = [g.packed() if isinstance(g, parallel.GatheredStrings) else g for g in rec.gathered]
            handles_per_unit = len(rec.gathered)
            assert step_gather.steps == steps // group and step_gather.gathers == step_gather.steps * handles_per_unit, (
                "every timed step must end with its gather", step_gather.steps, step_gather.gathers, steps, group)
            t = torch.tensor([elapsed], dtype=torch.float64, device=device)
            every = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(every, t)
            per_rank_s = [float(v.item()) for v in every]
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            collective = {"gathers": step_gather.gathers, "steps": step_gather.steps,
                          "collective_ms_per_step": round(step_gather.collective_ms_per_step() * step_gather.steps / steps, 4),
                          "slot_bytes": [int(v) for _, v in sorted(step_gather.caps.items())],
                          "slots_outgrown": step_gather.overflows,
                          "per_rank_s": [round(v, 6) for v in per_rank_s],
                          "note": "per unit of %d step(s): per encoder handle one all-gather of lengths + totals and one of "
                                  "the bytes in fixed slots (parallel.gather_encoded_async), on the unit's stream, nothing "
                                  "read back; ms between HIP events around them" % group}
        assert rec.x_hat.shape == x.shape
        # decode parity: what the decoder returned IS the quantised latent the encoder coded (every image)
        y_coded = rec.out[0].coder_inputs[0]
        y_decoded = rec.x_hat._tfc_keep[-1]
        want = model.entropy_model.quantize(y_coded) if workload == "bls2017" else torch.round(y_coded.float()).to(y_coded.dtype)
        assert torch.equal(y_decoded, want), "decompress did not return the quantised latents"
        strings = rec.strings
        nbytes = sum(len(bytes(s)) for arr in strings for s in arr.reshape(-1))
        if gathered:
            # the last unit's first handle as every rank now has it: this rank's images sit at their place, byte for byte
            blob_all, offs_all = (t.cpu().numpy() for t in gathered[0])
            mine = strings[0].reshape(-1)
            assert len(offs_all) - 1 == world * len(mine)
            for i, s_i in enumerate(mine):
                k = rank * len(mine) + i
                assert bytes(blob_all[offs_all[k]:offs_all[k + 1]]) == bytes(s_i), "gathered strings differ from this rank's own"
        res = None
        if rank == 0:
            pixels = world * batch * hw[0] * hw[1]
            conv_ms = kern["conv2d"][0] / lone_steps
            flops = model_conv_flops(workload, batch, hw)
            conv_tflops = flops / 1e12 / (conv_ms / 1e3) if conv_ms > 0 else 0.0
            # dense matrix-core peak of the transforms' arithmetic type: bf16 MFMA, or v_mfma_f32_32x32x2_f32 (256 CUs x 256 flop/clk x 2.4 GHz)
            mfma_peak = 2500.0 if dtype_name == "bf16" else 157.3
            res = {
                "value": round(pixels / 1e6 / (elapsed / steps), 2), "unit": "Mpixels/s",
                "ms_per_step": round(1e3 * elapsed / steps, 3), "steps": steps, "warmup": warmup,
                "timed_passes_ms_per_step": [round(1e3 * v / steps, 3) for v in pass_s],
                "memory_gb": {"torch_reserved": round(torch.cuda.memory_reserved(device) / 2 ** 30, 1),
                              "torch_allocated_peak": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 1),
                              "library_cached": round(pipeline.cached_bytes() / 2 ** 30, 1),
                              "device_free": round(torch.cuda.mem_get_info(device)[0] / 2 ** 30, 1),
                              "torch_alloc_retries": torch.cuda.memory_stats(device).get("num_alloc_retries", 0)},
                "workload": f"{workload} compress+decompress, {batch} images of {hw[1]}x{hw[0]} per GPU, 192 filters, "
                            f"random-init weights" + (", hyperprior calibrated (scale_index_histogram)" if hist is not None else ""),
                "dtype": dtype_name,
                "steps_in_flight": len(lanes) * group,
                "streams": len(step_lanes.lanes),
                "steps_per_coder_launch": group,
                "strings_fetched_to_host_in_timed_region": True,
                "collective": collective,
                "lone_step": {"ms_per_step": round(1e3 * lone_s / lone_steps, 3),
                              "mpixels_s": round(batch * hw[0] * hw[1] / 1e6 / (lone_s / lone_steps), 2),
                              "kernels_ms": {"conv2d": round(conv_ms, 3),
                                             "coder_encode": round(kern["enc_kernel"][0] / lone_steps, 3),
                                             "coder_decode": round(kern["dec_kernel"][0] / lone_steps, 3),
                                             "gdn_forward": round(kern["gdn_forward"][0] / lone_steps, 3)},
                              "note": "one step at a time, whole chip, nothing else resident: where the kernel split is measured"},
                "bits_per_pixel": round(8.0 * nbytes / (batch * hw[0] * hw[1]), 4),
                "roofline": {"bound": "mfma", "kernel": "conv2d (all SignalConv2D launches of a step)",
                             "achieved": round(conv_tflops, 1), "peak": mfma_peak, "unit": "TFLOP/s",
                             "frac": round(conv_tflops / mfma_peak, 4), "traffic": None,
                             "algorithmic_flops": int(flops),
                             **({"note": "float32 layers of >= 16 channels run on the bfloat16 matrix cores: three bf16 planes per "
                                         "operand, six bf16 MFMA products per float32 product (csrc/signal_conv.hip conv_split_x_kernel; "
                                         "TFC_CONV_F32=native: the float32 MFMA kernel) — `achieved` counts the float32 problem's flops, "
                                         "`peak` is the float32 MFMA's; the six-product ceiling is 2500 / 6 = 417 TFLOP/s"}
                                if dtype_name != "bf16" else {})},
            }
            if hist is not None:
                res["scale_index_histogram"] = [int(v) for v in hist]
            if cpu and world == 1:
                res["cpu_baseline"] = model_cpu_baseline(model, rec.out[0], strings[0], hw)
                assert res["cpu_baseline"]["bytes_identical_to_gpu"], (
                    "GPU strings differ from the CPU reference's: %s" % json.dumps(res["cpu_baseline"]))
                if workload == "bmshj2018":
                    res["cpu_baseline"]["side_strings"] = side_strings_check(model, rec.out[1], strings[1])
                    assert res["cpu_baseline"]["side_strings"]["bytes_identical_to_gpu"], (
                        "GPU side strings differ from the CPU reference's: %s" % json.dumps(res["cpu_baseline"]["side_strings"]))
        del rec
    return res


def conv_layer_table(device, batch=128):
    """SignalConv2D per layer shape of BASELINE config 4 (bmshj2018, `batch` images of 768x512, bf16): kernel
    time between HIP events, 2 M K N FLOP (synthesis on its output grid), fraction of the 2.5 PFLOP/s dense
    bf16 MFMA peak."""
    from compression_amd.layers import conv2d_down, conv2d_up
    C_, dt, H, W = 192, torch.bfloat16, 512, 768
    gen = torch.Generator(device="cpu").manual_seed(5)
    k = lambda kh, ci, co: (torch.randn(kh, kh, ci, co, generator=gen) / (kh * kh * ci) ** 0.5).to(device)
    layers = [
        ("analysis 5x5 3->192 /2", conv2d_down, (H, W, 3), 5, 3, C_, 2, False),
        ("analysis 5x5 192->192 /2 @384x256", conv2d_down, (H // 2, W // 2, C_), 5, C_, C_, 2, False),
        ("analysis 5x5 192->192 /2 @192x128", conv2d_down, (H // 4, W // 4, C_), 5, C_, C_, 2, False),
        ("analysis 5x5 192->192 /2 @96x64", conv2d_down, (H // 8, W // 8, C_), 5, C_, C_, 2, False),
        ("synthesis 5x5 192->192 x2 @48x32", conv2d_up, (H // 16, W // 16, C_), 5, C_, C_, 2, True),
        ("synthesis 5x5 192->192 x2 @96x64", conv2d_up, (H // 8, W // 8, C_), 5, C_, C_, 2, True),
        ("synthesis 5x5 192->192 x2 @192x128", conv2d_up, (H // 4, W // 4, C_), 5, C_, C_, 2, True),
        ("synthesis 5x5 192->3 x2 @384x256", conv2d_up, (H // 2, W // 2, C_), 5, C_, 3, 2, True),
    ]
    rows = []
    for name, fn, shp, kh, ci, co, stride, up in layers:
        x = torch.randn((batch,) + shp, device=device).to(dt)
        w = k(kh, ci, co)
        b = torch.zeros(co, device=device)
        for _ in range(2):
            fn(x, w, b, stride)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        e0.record()
        for _ in range(reps):
            fn(x, w, b, stride)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out_pix = batch * shp[0] * shp[1] * (stride * stride if up else 1.0 / (stride * stride))
        flops = 2.0 * out_pix * kh * kh * ci * co / (stride * stride if up else 1)
        tf = flops / 1e12 / (ms / 1e3)
        rows.append({"layer": name, "ms": round(ms, 3), "tflops": round(tf, 1), "frac_of_2500": round(tf / 2500.0, 4)})
        del x
    # the two big layers as a model step launches them: GDN / IGDN as the layer's activation, inside the kernel
    # (tfc_conv2d_gdn; FLOP count: the convolution's + the 2 C^2 per pixel of the norm's contraction)
    from compression_amd.layers.functional import GDNPrepared, conv2d_gdn
    prepared = GDNPrepared(torch.rand(C_) + 1.0, torch.rand(C_, C_) * 0.01 + 0.1 * torch.eye(C_), dt)
    for name, shp, up in (("analysis 5x5 192->192 /2 @384x256 + GDN", (H // 2, W // 2, C_), False),
                          ("synthesis 5x5 192->192 x2 @192x128 + IGDN", (H // 4, W // 4, C_), True)):
        x = torch.randn((batch,) + shp, device=device).to(dt)
        w = k(5, C_, C_)
        b = torch.zeros(C_, device=device)
        fused = True
        for _ in range(2):
            _, fused = conv2d_gdn(x, w, b, 2, up, prepared, up)
        if not fused:
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            conv2d_gdn(x, w, b, 2, up, prepared, up)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 3
        out_pix = batch * shp[0] * shp[1] * (4 if up else 0.25)
        flops = 2.0 * out_pix * 25 * C_ * C_ / (4 if up else 1) + 2.0 * out_pix * C_ * C_
        tf = flops / 1e12 / (ms / 1e3)
        rows.append({"layer": name, "ms": round(ms, 3), "tflops": round(tf, 1), "frac_of_2500": round(tf / 2500.0, 4)})
        del x
    return {"workload": f"SignalConv2D layer shapes of bmshj2018 at {batch} x 768x512, bf16", "layers": rows}


def training_figures(device):
    """SURVEY 8(f) row 2, the training-time path: one bls2017 training step (forward, backward through every HIP kernel,
    Adam) at 16 x 256x256 bf16 — wall time per step and the library's kernel split (tfc_profile_enable) — and the
    weight-gradient kernel alone on the 5x5 / 2 192 -> 192 layer at 16 x 384x256."""
    from compression_amd import models
    from compression_amd.layers.functional import conv2d_wgrad
    torch.manual_seed(0)
    model = models.BLS2017Model(lmbda=0.01, num_filters=192, compute_dtype=torch.bfloat16).to(device)
    x = torch.from_numpy(synthetic.lowpass_images(8, 256, 256, seed=3)).to(device).repeat(2, 1, 1, 1)
    model(x)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)

    def step():
        opt.zero_grad()
        loss, _, _ = model(x, training=True)
        loss.backward()
        opt.step()
        return loss

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n
    _lib.lib().tfc_profile_enable(1)
    step()
    torch.cuda.synchronize()
    split = {}
    for name in ("conv2d", "conv2d_wgrad", "gdn_forward", "gdn_backward_fused", "gdn_backward_params",
                 "factorized_forward", "factorized_backward"):
        ms, cnt = profile_query(name)
        if cnt:
            split[name] = round(ms, 3)
    _lib.lib().tfc_profile_enable(0)
    del model, opt
    gen = torch.Generator(device="cpu").manual_seed(1)
    a = torch.randn(16, 256, 384, 192, generator=gen).to(torch.bfloat16).to(device)
    b = torch.randn(16, 128, 192, 192, generator=gen).to(torch.bfloat16).to(device)
    conv2d_wgrad(a, b, (5, 5), 2, False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        conv2d_wgrad(a, b, (5, 5), 2, False)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / 5
    flops = 2.0 * 16 * 128 * 192 * 25 * 192 * 192
    return {"workload": "bls2017 training step (forward + backward + Adam), 16 x 256x256, bf16, synthetic images",
            "ms_per_step": round(1e3 * wall, 3), "mpixels_s": round(16 * 256 * 256 / 1e6 / wall, 1),
            "kernels_ms": split, "note": "wall time of a host-driven step; kernels_ms = the library's launches of one step",
            "wgrad_5x5_s2_192_192_at_16x384x256": {"ms": round(ms, 3), "tflops": round(flops / 1e12 / (ms / 1e3), 1),
                                                   "frac_of_2500": round(flops / 1e12 / (ms / 1e3) / 2500.0, 4)}}


def model_group(args, workload):
    return args.model_group if args.model_group > 0 else (8 if workload == "bmshj2018" else 1)


def model_workload(args, world, rank, device, distributed):
    """`--workload bls2017|bmshj2018`: the model step as the headline line (BASELINE configs 1/4/5)."""
    import torch.distributed as dist
    from compression_amd import pipeline
    res = model_bench(args.workload, args.model_dtype, device, batch=args.batch, steps=args.steps,
                      warmup=args.warmup, lanes=pipeline.StepLanes(args.model_depth or 6, device),
                      cpu=not args.no_cpu_baseline, rank=rank, world=world, distributed=distributed,
                      group=model_group(args, args.workload), queue=args.model_queue)
    if rank == 0:
        line = {
            "metric": "Mpixels/s encode+decode round-trip (bit-exact)",
            "value": res["value"], "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.model_dtype, "data": "synthetic",
            "config": {"workload": res["workload"], "parallelism": f"batch-sharded x{world}",
                       "steps_in_flight": res["steps_in_flight"], "steps_per_coder_launch": res["steps_per_coder_launch"],
                       "collectives": "broadcast of weights + tables at setup; every step ends, inside the timed region, "
                                      "with the all-gather of its strings' lengths and of their bytes in fixed slots "
                                      "(RCCL, on the step's stream, nothing read back)" if distributed else "none"},
        }
        if distributed and res.get("collective"):
            pixels_rank = res["value"] * 1e6 * (res["ms_per_step"] / 1e3) / world      # pixels of one rank's step
            line["per_rank"] = [{"rank": r, "mpixels_s": round(pixels_rank * res["steps"] / 1e6 / max(sec, 1e-9), 2)}
                                for r, sec in enumerate(res["collective"]["per_rank_s"])]
            line["collective_ms_per_step"] = res["collective"]["collective_ms_per_step"]
        for key in ("bits_per_pixel", "lone_step", "roofline", "scale_index_histogram", "cpu_baseline", "collective"):
            if key in res:
                line[key] = res[key]
        print(json.dumps(line))
    if distributed:
        dist.destroy_process_group()


def c2_run(args, lookup, lookup_t, device, world, rank, distributed, escape_fraction, steps, inflight,
           serial=True, fold=False, to_host=False, repeats=1):
    """The coder round trip at BASELINE config 2 with `escape_fraction` of the symbols out of range: exactly
    `steps` steps, `inflight` per launch group.  Returns the measurements (every rank) — rank 0 formats."""
    import hashlib
    if distributed:
        import torch.distributed as dist
    inflight = max(1, min(inflight, steps))
    # every slot in flight codes its own tensor (inputs resident in HBM)
    slots = [sample_symbols_device(lookup, 1000 * rank + k, device, escape_fraction, fold) for k in range(inflight)]
    # one stream: the launches of a group fill the chip on their own (a decoder workgroup takes a whole
    # CU's LDS), groups on different streams would only queue behind each other's workgroups
    side_streams = [torch.cuda.Stream(device=device)]
    sink = HostSink(device, inflight) if to_host else None
    torch.cuda.synchronize()

    def run_steps(total_steps, depth, mode):
        """Exactly `total_steps` steps, `depth` in flight, enqueued by this one thread; returns
        (seconds, per-step results)."""
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()
        results = []
        t0 = time.perf_counter()
        if depth <= 1:
            for k in range(total_steps):
                results.append(one_step(lookup_t, slots[0], mode) + (0,))
                torch.cuda.current_stream().synchronize()
        else:
            # groups of `depth` steps, one launch per group and direction
            for g, k0 in enumerate(range(0, total_steps, depth)):
                idx = [k % depth for k in range(k0, min(k0 + depth, total_steps))]
                with torch.cuda.stream(side_streams[g % len(side_streams)]):
                    for r, slot in zip(step_group(lookup_t, [slots[i] for i in idx], mode, sink), idx):
                        results.append(r + (slot,))
        t_enqueued = time.perf_counter() - t0
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()
        return time.perf_counter() - t0, results, t_enqueued

    def verify(results):
        """Parity gate (outside the timed region): every step's round trip is exact, the sanity flags
        are true, and no deferred error is pending."""
        for h, d, dec_r, ok_r, slot in results:
            tfc.entropy_decode_status(d)
            tfc.entropy_encode_status(h)
            assert bool(ok_r.all()), "EntropyDecodeFinalize reported a failed stream"
            assert torch.equal(dec_r.reshape(STREAMS, ELEMS), slots[slot]), "decode(encode(x)) != x"

    flight_mode = "throughput" if inflight > 1 else "latency"
    _, res, _ = run_steps(max(min(args.warmup, 3), 1), 1, "latency")
    verify(res)
    if inflight > 1:
        # the timed pattern once, untimed: primes both streams and the stream-ordered memory pool with
        # exactly the buffers the timed region asks for (a fresh 50-100 MB driver allocation per buffer
        # would otherwise be timed instead of the coder)
        _, res, _ = run_steps(steps, inflight, flight_mode)
        verify(res)
    del res
    torch.cuda.synchronize()

    m = {"inflight": inflight, "steps": steps, "escape_fraction": escape_fraction, "escape_share": escape_share(lookup, slots[0])}
    if serial:
        # one batch at a time (config 2 as literally written), per-kernel durations with the GPU to one launch
        # (a counter-profiling run leaves the single-step lane launches out: per-launch counter averages of the
        # lane kernels must be over launches that carry the same number of steps)
        profiling = bool(os.environ.get("TFC_PROFILE_STEPS_PER_LAUNCH"))
        for mode, key in (("latency", "serial"), ("throughput", "serial_lanes")):
            if profiling and key == "serial_lanes":
                m[key] = None
                continue
            _lib.lib().tfc_profile_enable(1)
            n = min(steps, 5 if mode == "latency" else 2)
            sec, results, _ = run_steps(n, 1, mode)
            enc_ms, enc_n = profile_query("enc_kernel")
            dec_ms, dec_n = profile_query("dec_kernel")
            _lib.lib().tfc_profile_enable(0)
            verify(results)
            del results
            m[key] = {"seconds": sec, "steps": n, "enc_ms": enc_ms / max(enc_n, 1), "dec_ms": dec_ms / max(dec_n, 1)}
    # the timed region: exactly `steps` steps
    # `repeats` times over (median of >= 5 runs), each bracketed by its own barrier + synchronize; the
    # median counts, every repetition's time is printed
    _lib.lib().tfc_profile_enable(1)
    reps, reps_local = [], []
    results = None
    for _ in range(max(1, repeats)):
        if results is not None:
            verify(results)
            del results
        e, results, t_enqueued = run_steps(steps, inflight, flight_mode)
        reps_local.append(e)
        if distributed:                         # every region's time is the slowest rank's
            t = torch.tensor([e], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e = float(t[0].item())
        reps.append(e)
    elapsed = sorted(reps)[len(reps) // 2]
    elapsed_local = sorted(reps_local)[len(reps_local) // 2]
    cenc_ms, cenc_n = profile_query("enc_kernel")
    cdec_ms, cdec_n = profile_query("dec_kernel")
    # the pipelined lane kernels (csrc/range_pipe.h): the launches inside an encode / decode call, each timed on its own
    stages = {}
    for name in ("enc_expand", "enc_chain", "dec_chain", "dec_parse_next", "dec_parse"):
        ms, cnt = profile_query(name)
        if cnt:
            stages[name] = ms / cnt
    _lib.lib().tfc_profile_enable(0)
    verify(results)
    # every distinct input's bytes (sha256 of the packed blob and of its offsets), for the comparison with the CPU reference
    shas = {}
    for r in results:
        if r[4] not in shas:
            shas[r[4]] = handle_hashes(r[0])
    del results
    if sink is not None:
        m["host_bytes"] = sink.bytes
    m.update(stages=stages, elapsed=elapsed, elapsed_local=elapsed_local, repetitions=reps, t_enqueued=t_enqueued, enc_tr=cenc_ms / max(cenc_n, 1), dec_tr=cdec_ms / max(cdec_n, 1),
             total_bytes=shas[0][2], blob_sha=shas[0][0], offs_sha=shas[0][1], slot0=slots[0],
             slots=slots, slot_shas=[shas[k] for k in sorted(shas)])
    return m


def saturation_curve(lookup, lookup_t, device, points, bytes_per_batch, cpu_value=None, distinct=32):
    """Batches per launch group -> Mpixels/s: the same round trip as the headline (throughput-mode handles, one group =
    one encode call + one decode call for all its batches, nothing read back), for 1 ... 128 batches of 512 streams per
    group.  A stream is a strict chain, so a group takes (symbols per stream) x (cycles per chain step) whatever its
    size until every SIMD holds a chain wave; past that the chip-wide streaming kernels (expansion, parse) bound it.
    Inputs: `distinct` differently seeded tensors used cyclically (each 100 MB, far beyond L2 / MALL).  Per point: rate,
    the time of a group, per-kernel times (HIP events of the library around each kernel, summed over the group's
    launches), the kernel that takes longest and its algorithmic HBM rate."""
    top = max(points)
    slots = [sample_symbols_device(lookup, 7000 + k, device) for k in range(min(top, distinct))]
    symbols = STREAMS * ELEMS
    alg_dir = 4 * symbols + bytes_per_batch            # one direction of one batch: symbols in + code bytes out (or the mirror)
    rows = []
    for nb in points:
        values = [slots[k % len(slots)] for k in range(nb)]
        res = step_group(lookup_t, values, "throughput")                 # untimed: primes the pools with this group's buffers
        torch.cuda.synchronize()
        for (h, d, dec_r, ok_r), v in zip(res, values):
            assert bool(ok_r.all()) and torch.equal(dec_r.reshape(STREAMS, ELEMS), v), "decode(encode(x)) != x"
        del res
        reps = max(3, min(6, 120 // nb))
        _lib.lib().tfc_profile_enable(1)
        torch.cuda.synchronize()
        secs = []
        for _ in range(reps):
            t0 = time.perf_counter()
            res = step_group(lookup_t, values, "throughput")
            del res                                                       # handles and tensors go back in stream order
            torch.cuda.synchronize()
            secs.append(time.perf_counter() - t0)
        # the median group (a group that met a driver allocation — seen once at the 64-batch point: 63 ms against 23 — is
        # an outlier, not the point's rate)
        sec = sorted(secs)[len(secs) // 2]
        stages, launches = {}, {}
        for name in ("enc_expand", "enc_chain", "dec_chain", "dec_parse_next", "dec_parse", "enc_kernel", "dec_kernel"):
            ms, cnt = profile_query(name)
            if cnt:
                stages[name] = ms / reps
                launches[name] = cnt / reps
        _lib.lib().tfc_profile_enable(0)
        # the kernel that takes longest (dec_parse_next rides beside dec_chain: its span is the chain's)
        kernels = {k: v for k, v in stages.items() if k not in ("enc_kernel", "dec_kernel", "dec_parse_next")}
        dom = max(kernels, key=kernels.get) if kernels else None
        mpix = nb * STREAMS * PIXELS_PER_STREAM / 1e6 / sec
        row = {"batches_per_launch": nb, "streams": nb * STREAMS, "mpixels_s": round(mpix, 1),
               "ms_per_group": round(1e3 * sec, 3), "ms_per_batch": round(1e3 * sec / nb, 4),
               "groups_ms": [round(1e3 * v, 3) for v in secs],
               "encode_call_ms": round(stages.get("enc_kernel", 0.0), 3), "decode_call_ms": round(stages.get("dec_kernel", 0.0), 3),
               "kernels_ms": {k: round(v, 3) for k, v in kernels.items()},
               "launches_per_direction": round(launches.get("enc_kernel", 1)),
               "dominant_kernel": (dom + "_kernel") if dom else None,
               "dominant_kernel_algorithmic_gbs": round(nb * alg_dir / 1e9 / (kernels[dom] / 1e3), 1) if dom else None,
               "path_algorithmic_gbs": round(2 * nb * alg_dir / 1e9 / sec, 1),
               "path_frac_of_hbm_peak": round(2 * nb * alg_dir / 1e9 / sec / HBM_PEAK_GBS, 4)}
        if cpu_value:
            row["speedup_vs_cpu_baseline"] = round(mpix / cpu_value, 2)
        rows.append(row)
        torch.cuda.empty_cache()
    del slots
    return {"note": "round trip of N batches (N x 512 streams x 49152 symbols) as ONE group: one encode call + one decode call "
                    "(tfc_encoder_encode_many / tfc_decoder_decode_many; a call is split into several launches where the "
                    "pipelined kernels' temporaries would pass 6 GB), throughput-mode handles, strings stay in HBM; "
                    f"{len(rows)} points, each the median of 3-6 groups (timed one by one) after one untimed group; inputs: "
                    f"{min(top, distinct)} differently seeded tensors used cyclically; every point's decode checked against its input",
            "algorithmic_bytes_per_batch_and_direction": int(alg_dir),
            "points": rows}


def handle_hashes(h):
    """(sha256 of the packed blob, sha256 of the int64 offsets, total bytes) of an encoder handle's strings — all its
    streams."""
    import hashlib
    tfc.entropy_encode_finalize(h)
    blob = h.blob.cpu().numpy()
    offs = h.offsets.cpu().numpy().astype(np.int64)
    return hashlib.sha256(blob.tobytes()).hexdigest(), hashlib.sha256(offs.tobytes()).hexdigest(), int(offs[-1])


def compare_slots_with_cpu(lib, lookup, slots, shas, threads):
    """Every distinct input of a run coded by the CPU reference, sha256 of blob and offsets against the GPU's: returns
    (slots compared, slots identical, bytes compared)."""
    import hashlib
    same = total = 0
    for value_t, (blob_sha, offs_sha, nbytes) in zip(slots, shas):
        _, cpu_blob, cpu_offs = lib.encode(lookup, value_t.cpu().numpy(), threads=threads)
        same += int(hashlib.sha256(np.ascontiguousarray(cpu_blob).tobytes()).hexdigest() == blob_sha and
                    hashlib.sha256(np.ascontiguousarray(cpu_offs, np.int64).tobytes()).hexdigest() == offs_sha)
        total += nbytes
    return len(shas), same, total


def escape_object(args, lookup, lookup_t, device, fraction, cpu_value, fold=False):
    """The --steps command again on another input law — `fraction` of the symbols replaced by far-out values, or
    (`fold`) the overflow draws folded away: throughput, and slot 0's bytes against the CPU reference coder's."""
    m = c2_run(args, lookup, lookup_t, device, 1, 0, False, fraction, args.steps, args.inflight, serial=False, fold=fold)
    value = STREAMS * PIXELS_PER_STREAM / 1e6 / (m["elapsed"] / args.steps)
    obj = {"law": "overflow draws folded onto the neighbouring symbol (no escape codes)" if fold else
                  f"8(d) law + {fraction:g} of the symbols replaced by +-(len + Geometric(0.2))",
           "symbols_on_the_escape_path": round(m["escape_share"], 5),
           "value": round(value, 2), "unit": "Mpixels/s",
           "ms_per_step": round(1e3 * m["elapsed"] / args.steps, 4), "steps": args.steps,
           "steps_in_flight": m["inflight"],
           "bits_per_symbol": round(8.0 * m["total_bytes"] / (STREAMS * ELEMS), 4),
           "kernels_ms_in_flight": {"enc_kernel": round(m["enc_tr"], 4), "dec_kernel": round(m["dec_tr"], 4)}}
    if not args.no_cpu_baseline:
        from oracle import oracle
        import hashlib
        lib = oracle.best()
        _, cores, _ = usable_cores()
        value_h = m["slot0"].cpu().numpy()
        enc, dec, total, ok = lib.bench_roundtrip(lookup, value_h, threads=cores, reps=4)
        assert ok
        rt = float(np.median((enc + dec)[1:]))
        compared, identical, nbytes = compare_slots_with_cpu(lib, lookup, m["slots"], m["slot_shas"], cores)
        same = compared == identical
        cpu_mpix = STREAMS * PIXELS_PER_STREAM / 1e6 / rt
        obj["cpu_baseline"] = {"value": round(cpu_mpix, 2), "unit": "Mpixels/s", "cores": cores, "kind": lib.kind,
                               "sample": "the same 512-stream batch (slot 0), 4 repetitions (first discarded)"}
        obj["speedup_vs_cpu_baseline"] = round(value / cpu_mpix, 2)
        obj["bytes_identical_to_gpu"] = bool(same)
        obj["slots_compared"], obj["slots_identical"] = compared, identical
        assert same, "GPU bytes differ from the CPU reference's (escape run)"
    return obj


def free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def spawn_command(args_gpus, argv, port):
    """The command `python bench.py --gpus N ...` re-executes itself as when no launcher has set WORLD_SIZE: the
    driver's own multi-GPU form (one rank per GPU of one node, rendezvous on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def maybe_spawn(args):
    """`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process is not a rank, it STARTS the N ranks
    (and fails loudly when the node has fewer than N devices).  Returns True when it did (the ranks' output is this
    process's output, their exit status its own)."""
    if "WORLD_SIZE" in os.environ or args.gpus <= 1:
        return False
    import subprocess
    if not args.spawn_check:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} asked for, {have} HIP device(s) visible on this node")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    rc = subprocess.run(spawn_command(args.gpus, sys.argv[1:], free_port()), env=env).returncode
    if rc:
        raise SystemExit(rc)
    return True


def spawn_check(world, rank):
    """The ranks of `--spawn-check`: rendezvous over gloo (no device), count each other, rank 0 prints one line."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seen = torch.zeros(world, dtype=torch.int64)
    seen[rank] = 1
    dist.all_reduce(seen)
    if rank == 0:
        print(json.dumps({"spawn_check": True, "world": world, "ranks_seen": int(seen.sum()),
                          "launcher": "torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1"}))
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed regions of --steps steps each (barrier + synchronize around every one); the median is `value`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the headline measurement (no single_batch / escapes / models / conv sub-objects)")
    ap.add_argument("--inflight", type=int, default=20,
                    help="batches per launch group (one host thread; 1 = one batch at a time): the headline's operating point. "
                         "FIXED at 20 — `value` is the rate of groups of 20 batches whatever --steps is (fewer steps than "
                         "that: one smaller group); the `saturation` sub-object has the curve 1 ... 128")
    ap.add_argument("--escape-fraction", type=float, default=0.0)
    ap.add_argument("--workload", default="c2", choices=["c2", "bls2017", "bmshj2018"],
                    help="c2 (default, the headline): coder round trip; bls2017 / bmshj2018: "
                         "full model compress+decompress as the headline line (configs 1/4/5)")
    ap.add_argument("--batch", type=int, default=0, help="images per GPU for the model workloads")
    ap.add_argument("--model-dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--model-depth", type=int, default=0,
                    help="streams that carry model steps (1: one step at a time; 0: 6 — measured, profiles/r03_notes.md)")
    ap.add_argument("--model-group", type=int, default=0,
                    help="batches per coder launch (the models' compress_many / decompress_many: one launch per stage "
                         "of the pipelined lane kernels for all of them, a handful of waves whatever the group); "
                         "0: 8 for bmshj2018, 1 for bls2017 (its random-init tables' decoder image does not fit the LDS)")
    ap.add_argument("--model-queue", type=int, default=2, help="model steps enqueued per stream (--model-depth streams)")
    ap.add_argument("--model-steps", type=int, default=0,
                    help="timed steps of the `models` sub-objects (0: 32, and 128 where 8 batches share a coder launch)")
    ap.add_argument("--saturation", default="1,2,4,8,20,64,128",
                    help="batches per launch of the `saturation` sub-object (comma separated; empty: leave it out)")
    ap.add_argument("--spawn-check", action="store_true",
                    help="no measurement: start the --gpus ranks exactly as a measurement would, rendezvous over gloo on "
                         "the host, and print {world, ranks} (the CPU-tier test of the launch path)")
    ap.add_argument("--leg", default="all", choices=["all", "headline", "single_batch", "gdn"],
                    help="profiling aid: run ONE leg of the default line only and print a short line for it — `headline` "
                         "(the timed groups of --steps batches, throughput handles) or `single_batch` (BASELINE config 2 as "
                         "written: one batch at a time, latency handles) — so that a rocprofv3 summary of the command is a "
                         "summary of that leg's kernels (tools/r06_profiles.sh)")
    ap.add_argument("--collective-check", action="store_true",
                    help="diagnostic for a one-GPU box: run the single rank inside an RCCL process group of world size 1, "
                         "so that the model workloads' per-step gather (and every other distributed branch) executes")
    args = ap.parse_args()

    if maybe_spawn(args):
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.spawn_check:
        return spawn_check(world, rank)
    if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} needs HIP device {local_rank}; {torch.cuda.device_count()} visible")
    distributed = world > 1 or args.collective_check
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL's own stream at HIGH priority: HIP multiplexes the streams of one priority onto a few hardware queues, and a
        # collective that waits (by event) for one step's stream while sharing a hardware queue with another step's stream
        # holds that one up too (bls2017 with the per-step gather, world of 1: 10.0 ms per step against 5.9 without)
        opts = dist.ProcessGroupNCCL.Options()
        opts.is_high_priority_stream = True
        dist.init_process_group("nccl", device_id=device, pg_options=opts)

    if args.workload != "c2":
        return model_workload(args, world, rank, device, distributed)
    # the streams of the model sub-objects, before anything else creates streams in this process
    from compression_amd import pipeline
    step_lanes = pipeline.StepLanes(args.model_depth or 6, device) if (not args.no_extras and world == 1) else None

    lookup = build_tables(device)
    lookup_t = torch.from_numpy(lookup)            # tables are uploaded once (cached per tensor)
    extras = not args.no_extras and world == 1
    if args.leg == "gdn":
        if rank == 0:
            print(json.dumps({"leg": "gdn", "gdn_fwd": gdn_forward_bandwidth(device)}))
        if distributed:
            dist.destroy_process_group()
        return
    if args.leg != "all":
        single = args.leg == "single_batch"
        m = c2_run(args, lookup, lookup_t, device, world, rank, distributed, args.escape_fraction,
                   5 if single else args.steps, 1 if single else args.inflight, serial=False, repeats=args.repeats)
        if rank == 0:
            steps = m["steps"]
            print(json.dumps({"leg": args.leg, "metric": "Mpixels/s encode+decode round-trip (bit-exact)",
                              "value": round(world * STREAMS * PIXELS_PER_STREAM / 1e6 / (m["elapsed"] / steps), 2),
                              "unit": "Mpixels/s", "steps": steps, "steps_in_flight": m["inflight"],
                              "ms_per_step": round(1e3 * m["elapsed"] / steps, 4),
                              "timed_regions_ms_per_step": [round(1e3 * v / steps, 4) for v in m["repetitions"]],
                              "kernels_ms_in_flight": {"enc_kernel": round(m["enc_tr"], 4), "dec_kernel": round(m["dec_tr"], 4),
                                                       "stages": {k: round(v, 4) for k, v in (m.get("stages") or {}).items()}}}))
        if distributed:
            dist.destroy_process_group()
        return
    m = c2_run(args, lookup, lookup_t, device, world, rank, distributed, args.escape_fraction, args.steps,
               args.inflight, serial=True, repeats=args.repeats)
    inflight, elapsed, total_bytes = m["inflight"], m["elapsed"], m["total_bytes"]
    per_rank = None
    if distributed:
        # every rank's own rate and byte total, so that the N = 1 SCALE value can be checked against BENCH
        mine = torch.tensor([STREAMS * PIXELS_PER_STREAM / 1e6 / (m["elapsed_local"] / args.steps), float(total_bytes)],
                            dtype=torch.float64, device=device)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = {"mpixels_s": [round(float(t[0]), 2) for t in allr],
                    "slot0_bytes": [int(t[1]) for t in allr],
                    "note": "each rank's own rate over its own clock (value uses the slowest rank's) and the byte total "
                            "of its slot-0 batch; ranks code differently seeded batches"}

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        pixels_all = world * STREAMS * PIXELS_PER_STREAM
        value_mpix = pixels_all / 1e6 / (elapsed / args.steps)
        symbols = STREAMS * ELEMS
        enc_avg, dec_avg = m["serial"]["enc_ms"], m["serial"]["dec_ms"]     # a launch has the GPU to itself
        enc_tr, dec_tr = m["enc_tr"], m["dec_tr"]                           # timed region
        # algorithmic bytes per launch: encode reads 4 B/symbol and
        # writes the code bytes; decode reads the code bytes and writes 4 B/symbol.
        alg_dec = 4 * symbols + total_bytes
        alg_enc = 4 * symbols + total_bytes
        lanes = inflight > 1
        jobs_per_launch = min(inflight, 64) if lanes else 1      # steps coded by one launch of the kernel
        dom, dom_ms, dom_bytes = ("dec_kernel", dec_tr, alg_dec) if dec_tr >= enc_tr else (
            "enc_kernel", enc_tr, alg_enc)
        dom_bytes *= jobs_per_launch
        achieved = dom_bytes / 1e9 / (dom_ms / 1e3) if dom_ms > 0 else 0.0
        dom_symbol = {("dec_kernel", True): "dec_lanes_kernel", ("enc_kernel", True): "enc_lanes_kernel",
                      ("dec_kernel", False): "dec_fast_kernel", ("enc_kernel", False): "enc_fast_kernel"}[(dom, lanes)]
        stages = m.get("stages") or {}
        if lanes and stages:
            # throughput-mode calls run the pipelined kernels of csrc/range_pipe.h: the dominant KERNEL is one stage
            # (dec_parse_next runs beside dec_chain from its first released rows to its last: its span is the chain's)
            stage = max((k for k in stages if k != "dec_parse_next"), key=stages.get)
            dom_symbol, dom_ms = stage + "_kernel", stages[stage]
            dom_bytes = (alg_dec if stage.startswith("dec") else alg_enc) * jobs_per_launch
            achieved = dom_bytes / 1e9 / (dom_ms / 1e3)
        ser, ser_l = m["serial"], m["serial_lanes"]
        out = {
            "metric": "Mpixels/s encode+decode round-trip (bit-exact)",
            "value": round(value_mpix, 2),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "timed_regions_ms_per_step": [round(1e3 * v / args.steps, 4) for v in m["repetitions"]],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {
                "workload": "range_encode+range_decode, 512 streams x 49152 int32 latents "
                            "(192ch x 16 x 16 = one 256x256 image each), 192 Gaussian tables, "
                            "precision 12, escape coding enabled, per GPU",
                "streams_per_gpu": STREAMS, "symbols_per_stream": ELEMS,
                "input_law": "SURVEY 8(d): uniform 12-bit draws inverted through each row's CDF, overflow bucket included "
                             "(the rows' tail mass 2^-8 takes the Elias-gamma escape path)" +
                             (f"; {args.escape_fraction:g} of the symbols replaced by far-out values" if args.escape_fraction else ""),
                "symbols_on_the_escape_path": round(m["escape_share"], 5),
                "parallelism": f"batch-sharded x{world}",
                "steps_in_flight": inflight,
                "launch": "the coding calls of the steps in flight are one launch per direction "
                          "(tfc_encoder_encode_many / tfc_decoder_decode_many); every step keeps its own handles and strings",
                "host_threads": 1,
                "timing": f"{len(m['repetitions'])} timed regions of exactly --steps steps, each between its own barrier + "
                          "synchronize; value / ms_per_step = their median, timed_regions_ms_per_step = all of them",
                "strings": "stay in HBM inside their handles in the timed region (device finalize; the decoders read them "
                           "in place); sub-object strings_to_host: the same command with every string copied to pinned "
                           "host memory inside the timed region",
                "distinct_inputs": inflight,
                "library_mode": "TFC_MODE_THROUGHPUT handles (one code stream per lane), deferred errors, "
                                "device finalize" if lanes else "TFC_MODE_LATENCY handles (one wave per stream)",
                "operating_point": f"{inflight} batches (= {inflight * STREAMS} streams, {inflight * STREAMS // 64} chain waves) per "
                                   "launch group — fixed by --inflight (default 20), NOT by --steps: more steps are more "
                                   "groups of the same size.  A stream is a strict chain, so a group takes the same time for "
                                   "1 ... ~100 batches; `saturation` gives the rate at 1, 2, 4, 8, 20, 64, 128 batches per "
                                   "group and `single_batch` BASELINE config 2 as literally written (one batch at a time)",
            },
            "bits_per_pixel": round(8.0 * total_bytes / (STREAMS * PIXELS_PER_STREAM), 5),
            "bits_per_symbol": round(8.0 * total_bytes / symbols, 4),
            "gsymbols_per_s_roundtrip": round(world * symbols / 1e9 / (elapsed / args.steps), 3),
            "host_enqueue_ms_per_step": round(1e3 * m["t_enqueued"] / args.steps, 4),
            "kernels_ms": {"enc_kernel": round(enc_avg, 4), "dec_kernel": round(dec_avg, 4),
                           "note": "latency-mode kernels, one launch at a time"},
            "kernels_ms_in_flight": {"enc_kernel": round(enc_tr, 4), "dec_kernel": round(dec_tr, 4),
                                     "stages": {k: round(v, 4) for k, v in stages.items()},
                                     "note": "enc_kernel / dec_kernel: all launches of an encode / decode call of the group; "
                                             "stages: its kernels (csrc/range_pipe.h) — enc_chain runs beside enc_expand and "
                                             "dec_parse_next beside dec_chain on a second stream, dec_parse is the pass behind the chain"},
            "single_batch": {
                "note": "BASELINE config 2 as literally written: ONE 512-stream batch at a time, host waits for every step",
                "latency_mode": {"ms_per_step": round(1e3 * ser["seconds"] / ser["steps"], 4),
                                 "mpixels_s": round(pixels_all / 1e6 / (ser["seconds"] / ser["steps"]), 2),
                                 "enc_kernel_ms": round(ser["enc_ms"], 4), "dec_kernel_ms": round(ser["dec_ms"], 4),
                                 "kernels": "one wave per stream"},
                "throughput_mode": None if ser_l is None else {
                    "ms_per_step": round(1e3 * ser_l["seconds"] / ser_l["steps"], 4),
                    "mpixels_s": round(pixels_all / 1e6 / (ser_l["seconds"] / ser_l["steps"]), 2),
                    "enc_kernel_ms": round(ser_l["enc_ms"], 4), "dec_kernel_ms": round(ser_l["dec_ms"], 4),
                    "kernels": "one lane per stream"},
            },
            "serial": {"ms_per_step": round(1e3 * ser["seconds"] / ser["steps"], 4),
                       "mpixels_s": round(pixels_all / 1e6 / (ser["seconds"] / ser["steps"]), 2),
                       "steps": ser["steps"], "library_mode": "TFC_MODE_LATENCY"},
            "roofline": {
                "bound": "hbm", "kernel": dom_symbol, "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": pmc_traffic(dom_symbol, PMC_PROFILE, CODER_SOURCES, jobs_per_launch),
                "traffic_source": f"stored: profiles/{PMC_PROFILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                  "this command on these kernel sources, taken by the builder — not measured in this run; "
                                  "2*FETCH + WRITE, KiB -> bytes); null = taken on other sources",
                "algorithmic_bytes": int(dom_bytes),
                "steps_per_launch": jobs_per_launch,
                "note": "serial chain per stream: bound by the latency of a step's instruction chain at one or two "
                        "waves per SIMD (a step = 64 streams of one wave), not by HBM; per-launch duration = "
                        "HIP-event average over the timed region; the aggregate rate of both directions is "
                        "path_gbytes_s_in_flight; see DESIGN.md",
                "path_gbytes_s_in_flight": round((alg_enc + alg_dec) * args.steps / 1e9 / elapsed, 2),
            },
        }
        if per_rank:
            out["per_rank"] = per_rank
        out["valu_issue_bound"] = valu_issue_floor(
            1e3 * elapsed / args.steps, ("enc_chain_kernel", "dec_chain_kernel") if lanes else ("enc_fast_kernel", "dec_fast_kernel"),
            jobs_per_launch)
        if world == 1:
            out["gdn_fwd"] = gdn_forward_bandwidth(device)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(lookup, m["slots"], m["slot_shas"])
            out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 2)
            out["single_batch"]["latency_mode"]["speedup_vs_cpu_baseline"] = round(
                out["single_batch"]["latency_mode"]["mpixels_s"] / out["cpu_baseline"]["value"], 2)
            if out["single_batch"]["throughput_mode"]:
                out["single_batch"]["throughput_mode"]["speedup_vs_cpu_baseline"] = round(
                    out["single_batch"]["throughput_mode"]["mpixels_s"] / out["cpu_baseline"]["value"], 2)
            assert out["cpu_baseline"]["bytes_identical_to_gpu"], "GPU bytes differ from the CPU reference's"
        lat = out["single_batch"]["latency_mode"]
        out["config"]["single_batch"] = (
            f"BASELINE config 2 as literally written — ONE 512-stream batch at a time, host waits for every step: "
            f"{lat['mpixels_s']} Mpixels/s ({lat['ms_per_step']} ms per batch"
            + (f", {lat['speedup_vs_cpu_baseline']}x the CPU reference" if "speedup_vs_cpu_baseline" in lat else "")
            + f"); the headline `value` is {inflight} such batches per launch group")
        bytes_per_batch = int(total_bytes)
        del m
        if extras and args.saturation:
            points = sorted({int(v) for v in args.saturation.split(",") if v.strip()})
            out["saturation"] = saturation_curve(lookup, lookup_t, device, points, bytes_per_batch,
                                                 out.get("cpu_baseline", {}).get("value"))
            out["saturation"]["headline_point"] = inflight
            torch.cuda.empty_cache()
            pipeline.empty_cache()
        if extras:
            torch.set_num_threads(1)
            mh = c2_run(args, lookup, lookup_t, device, 1, 0, False, args.escape_fraction, args.steps, args.inflight,
                        serial=False, to_host=True)
            out["strings_to_host"] = {
                "value": round(STREAMS * PIXELS_PER_STREAM / 1e6 / (mh["elapsed"] / args.steps), 2), "unit": "Mpixels/s",
                "ms_per_step": round(1e3 * mh["elapsed"] / args.steps, 4),
                "host_mbytes_per_step": round(mh["host_bytes"] / 1e6 / max(1, args.steps + args.steps), 3),
                "note": "offsets, then exactly the bytes, device -> pinned host on a second stream while the decoders run; "
                        "the host thread waits once per group, for the encoder"}
            del mh
            if args.escape_fraction == 0.0:
                out["escapes"] = {
                    "0.01": escape_object(args, lookup, lookup_t, device, 0.01, out.get("cpu_baseline", {}).get("value")),
                    "escape_free": escape_object(args, lookup, lookup_t, device, 0.0,
                                                 out.get("cpu_baseline", {}).get("value"), fold=True)}
            torch.cuda.empty_cache()
            out["conv"] = conv_layer_table(device)
            torch.cuda.empty_cache()
            out["training"] = training_figures(device)
            torch.cuda.empty_cache()
            out["models"] = {}
            for name, key in (("bls2017", "c1"), ("bmshj2018", "c4")):
                torch.cuda.empty_cache()
                g = model_group(args, name)
                out["models"][key] = model_bench(name, args.model_dtype, device,
                                                 steps=args.model_steps or (32 if g == 1 else 16 * g),
                                                 warmup=2, lanes=step_lanes, cpu=not args.no_cpu_baseline,
                                                 group=g, queue=args.model_queue)
            if args.model_dtype == "bf16":
                # the reference's default policy: float32 transforms (v_mfma_f32_32x32x2_f32, first-generation kernel)
                torch.cuda.empty_cache()
                out["models"]["c1_f32"] = model_bench("bls2017", "f32", device, steps=8, warmup=1, lanes=step_lanes,
                                                      cpu=False, group=1, queue=args.model_queue)
                torch.cuda.empty_cache()
                g = model_group(args, "bmshj2018")
                # (two of the lanes — as fast as three, measured — a float32 step keeps ~25 GB of activations per stream
                # live and torch's per-stream pools reserve ~70 GB for them: with three, the allocator ran out and
                # retried inside the timed passes, memory_gb.torch_alloc_retries)
                few = type("Lanes", (), {"lanes": step_lanes.lanes[:2]})()
                out["models"]["c4_f32"] = model_bench("bmshj2018", "f32", device, steps=2 * g, warmup=1, lanes=few,
                                                      cpu=False, group=g, queue=1, inflight_warmup=3)
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
