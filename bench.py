#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): Mpixels/s of the entropy-coding round trip
(range encode + range decode, bit-exact) on MI355X, at BASELINE config 2:
512 code streams x (16*16*192 = 49152) int32 latents = 512 images of 256x256,
192 discretised-Gaussian tables of precision 12 (escape coding enabled).

One "step" = one full pass of the hot path over the batch, inputs resident in
HBM: CreateRangeEncoder -> EntropyEncodeChannel -> EntropyEncodeFinalize ->
CreateRangeDecoder -> EntropyDecodeChannel -> EntropyDecodeFinalize, all through
the C ABI (libtfc_hip.so).  Prints ONE JSON line on rank 0.

Steps are independent batches, so `--inflight D` (default: up to 16, at most the usable host cores - 4, balanced over the steps) of them are in flight at a
time, each on its own host thread and HIP stream: one 512-stream step only puts one wave on
half of the GPU's 1024 SIMDs and every wave spends a third of its time in un-hideable
scalar/vector synchronisation stalls, which co-resident waves of other steps fill.  The
timed region still contains EXACTLY `--steps` complete steps; `serial` in the output is
the same measurement with one step at a time (the per-kernel durations used for the
roofline line are taken there, where a launch has the GPU to itself).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: the batch dimension shards (independent code streams), so every rank
codes its own 512-stream shard (weak scaling); no data-path collective, only the
barrier + max-reduce of the timing.
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


def build_tables(device):
    """192 Gaussian tables through the PRODUCT table builder (HIP pmf_to_quantized_cdf)."""
    pmfs, _ = synthetic.gaussian_pmfs(num_tables=CHANNELS)
    cdfs = []
    for p in pmfs:
        c = tfc.pmf_to_quantized_cdf(torch.from_numpy(p).to(device), PRECISION)
        cdfs.append(c.cpu().numpy())
    return synthetic.assemble_lookup(cdfs, PRECISION, overflow=True)


def profile_query(name):
    ms = C.c_double()
    n = C.c_int64()
    _lib.lib().tfc_profile_query(name.encode(), C.byref(ms), C.byref(n))
    return ms.value, n.value


def one_step(lookup_t, value_t):
    h = tfc.create_range_encoder([STREAMS], lookup_t)
    h = tfc.entropy_encode_channel(h, value_t)
    blob, offsets = tfc.gen_ops._finalize_device(h)
    d = tfc.create_range_decoder((blob, offsets, (STREAMS,)), lookup_t)
    d, decoded = tfc.entropy_decode_channel(d, [ELEMS], torch.int32)
    ok = tfc.entropy_decode_finalize(d)
    return blob, offsets, decoded, ok


def gdn_forward_bandwidth(device, steps=20):
    """BASELINE config 3 (second half of the metric): GDN forward on
    256 x 192 x 32 x 32 bf16 (NHWC [262144, 192]); algorithmic bytes = read x + write y."""
    from compression_amd.layers import gdn_forward
    torch.manual_seed(3)
    C, M = 192, 256 * 32 * 32
    x = torch.randn(M, C, device=device).bfloat16()
    beta = 1 + 0.1 * torch.rand(C)
    gamma = 0.1 * torch.eye(C) + 0.01 * torch.rand(C, C)
    for _ in range(3):
        y = gdn_forward(x, beta, gamma)
    torch.cuda.synchronize()
    _lib.lib().tfc_profile_enable(1)
    for _ in range(steps):
        y = gdn_forward(x, beta, gamma)
    torch.cuda.synchronize()
    ms, n = profile_query("gdn_forward")
    _lib.lib().tfc_profile_enable(0)
    avg_ms = ms / max(n, 1)
    nbytes = 2 * x.numel() * x.element_size()
    gbs = nbytes / 1e9 / (avg_ms / 1e3)
    # backward: x, g in; dx out is the algorithmic minimum (3 tensors).  The fused kernel
    # (x, g -> T, dx) plus the parameter pass (x, T -> dgamma, dbeta) move 4 + 2 tensors.
    from compression_amd.layers import gdn_backward
    g = torch.randn(M, C, device=device).bfloat16()
    for _ in range(2):
        gdn_backward(x, g, beta, gamma)
    torch.cuda.synchronize()
    _lib.lib().tfc_profile_enable(1)
    for _ in range(steps):
        gdn_backward(x, g, beta, gamma)
    torch.cuda.synchronize()
    passes = {}
    for name in ("gdn_backward_fused", "gdn_backward_t", "gdn_backward_dx", "gdn_backward_params"):
        pms, pn = profile_query(name)
        if pn:
            passes[name] = round(pms / pn, 4)
    _lib.lib().tfc_profile_enable(0)
    bwd_ms = sum(passes.values())
    bwd_bytes = 3 * x.numel() * x.element_size()
    return {"workload": "GDN fwd, [262144, 192] bf16 (= 256x192x32x32), alpha=1, eps=1",
            "kernel_ms": round(avg_ms, 4), "algorithmic_bytes": nbytes,
            "achieved": round(gbs, 1), "unit": "GB/s", "peak": HBM_PEAK_GBS,
            "frac": round(gbs / HBM_PEAK_GBS, 4), "bound": "hbm",
            "traffic": pmc_traffic("gdn_fwd_bf16_kernel<6, 0, true>"),
            "backward": {"kernel_ms": round(bwd_ms, 4), "passes_ms": passes,
                         "algorithmic_bytes": bwd_bytes,
                         "achieved": round(bwd_bytes / 1e9 / (bwd_ms / 1e3), 1) if bwd_ms else None,
                         "unit": "GB/s"}}


PMC_FILE = os.path.join(ROOT, "profiles", "r01_o_pmc_traffic.json")


def pmc_traffic(kernel_substring):
    """HBM bytes per launch of a kernel from the committed rocprofv3 --pmc passes
    (tools/pmc_on_box.sh; same bench command).  Counter unit is KiB.  Corrections as
    MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE counts 128-B read requests as 64 B
    (x2; calibrated in the same passes on torch's fp32->bf16 copy of a 201 MB tensor: reported
    98322 KiB, true 196608 KiB), WRITE_SIZE is exact on that kernel's 98304 KiB output."""
    try:
        table = json.load(open(PMC_FILE))
    except OSError:
        return None
    for name, ctrs in table.items():
        if kernel_substring in name and "FETCH_SIZE" in ctrs and "WRITE_SIZE" in ctrs:
            return int((2.0 * ctrs["FETCH_SIZE"]["mean"] + ctrs["WRITE_SIZE"]["mean"]) * 1024)
    return None


SQ_FILES = [os.path.join(ROOT, "profiles", f) for f in ("r01_o_sq_inflight1.json", "r01_m_sq_inflight1.json")]


def valu_issue_floor(ms_per_step, throughput_mode):
    """What bounds the coder: VALU issue.  SQ_ACTIVE_INST_VALU (quad-cycles, summed over waves; committed
    rocprofv3 --pmc passes of this command, tools/sq_on_box.sh) of the encoder + decoder launches of one
    step, spread over all SIMDs, is the time a step needs if every SIMD issued vector instructions
    without a gap."""
    want = ("enc_quad_kernel" if throughput_mode else "enc_fast_kernel", "dec_fast_kernel")
    quads = {}
    for path in SQ_FILES:
        try:
            table = json.load(open(path))
        except OSError:
            continue
        for name, row in table.items():
            for key in want:
                if key in name and key not in quads and row.get("SQ_ACTIVE_INST_VALU"):
                    quads[key] = row["SQ_ACTIVE_INST_VALU"]
    if len(quads) != 2:
        return None
    simds, clock_hz = 256 * 4, 2.4e9
    floor_ms = 1e3 * 4.0 * sum(quads.values()) / simds / clock_hz
    return {"valu_quad_cycles_per_step": {k: int(v) for k, v in quads.items()},
            "simds": simds, "clock_ghz": 2.4, "floor_ms_per_step": round(floor_ms, 4),
            "frac": round(floor_ms / ms_per_step, 4),
            "source": "profiles/r01_o_sq_inflight1.json, r01_m_sq_inflight1.json (SQ_ACTIVE_INST_VALU, rocprofv3 --pmc)"}


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


def cpu_baseline(lookup, value, total_bytes_gpu):
    """Reference coder core (oracle/_ref) or its restatement on the host cores,
    sharded over streams like the reference's ThreadPool::ParallelFor.
    This is the ONLY place bench.py touches oracle/."""
    from oracle import oracle
    lib = oracle.best()
    hw_threads, cores, quota = usable_cores()
    # keep this process's own OpenMP / torch worker threads from spinning next to the pool
    torch.set_num_threads(1)
    time.sleep(1.0)
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
        "bytes_identical_to_gpu": bool(total == total_bytes_gpu),
    }


def model_workload(args, world, rank, device, distributed):
    """Full compress + decompress of a target model on synthetic images (BASELINE configs
    1/4/5); informational — the headline line is the c2 workload."""
    import torch.distributed as dist
    dtype = torch.bfloat16 if args.model_dtype == "bf16" else torch.float32
    torch.manual_seed(0)
    if args.workload == "bls2017":
        model = tfc.models.BLS2017Model(num_filters=192, compute_dtype=dtype)
        batch, hw = args.batch or 512, (256, 256)
    else:
        model = tfc.models.BMSHJ2018Model(num_filters=192, compute_dtype=dtype)
        batch, hw = args.batch or 128, (512, 768)
    model = model.to(device).init_compression()
    base = torch.from_numpy(synthetic.lowpass_images(8, hw[0], hw[1], seed=2 + rank)).to(device)
    x = base.repeat((batch + 7) // 8, 1, 1, 1)[:batch].contiguous()

    def step():
        out = model.compress(x)
        return out, model.decompress(*out)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, x_hat = step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert x_hat.shape == x.shape
    nbytes = sum(len(bytes(s)) for arr in out if isinstance(arr, np.ndarray) for s in arr.reshape(-1))
    if rank == 0:
        pixels = world * batch * hw[0] * hw[1]
        print(json.dumps({
            "metric": "Mpixels/s encode+decode round-trip (bit-exact)",
            "value": round(pixels / 1e6 / (elapsed / args.steps), 2), "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.model_dtype, "data": "synthetic",
            "config": {"workload": f"{args.workload} compress+decompress, {batch} images of "
                                   f"{hw[1]}x{hw[0]} per GPU, 192 filters, random-init weights",
                       "parallelism": f"batch-sharded x{world}"},
            "bits_per_pixel": round(8.0 * nbytes / (batch * hw[0] * hw[1]), 4),
        }))
    if distributed:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inflight", type=int, default=0,
                    help="independent steps in flight (host threads x HIP streams); 1 = serial, "
                         "0 = up to 16, balanced over --steps")
    ap.add_argument("--escape-fraction", type=float, default=0.0)
    ap.add_argument("--workload", default="c2", choices=["c2", "bls2017", "bmshj2018"],
                    help="c2 (default, the headline): coder round trip; bls2017 / bmshj2018: "
                         "full model compress+decompress (informational)")
    ap.add_argument("--batch", type=int, default=0, help="images per GPU for the model workloads")
    ap.add_argument("--model-dtype", default="bf16", choices=["bf16", "f32"])
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    if args.workload != "c2":
        return model_workload(args, world, rank, device, distributed)

    lookup = build_tables(device)
    value = synthetic.sample_symbols(lookup, STREAMS, ELEMS, seed=rank,
                                     escape_fraction=args.escape_fraction, escape_seed=1000 + rank)
    lookup_t = torch.from_numpy(lookup)            # tables are uploaded once (cached per tensor)
    value_t = torch.from_numpy(value).to(device)   # inputs resident in HBM

    import threading
    from concurrent.futures import ThreadPoolExecutor

    def run_steps(total_steps, inflight):
        """Exactly `total_steps` steps, `inflight` at a time; returns (seconds, last results)."""
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        if inflight <= 1:
            for _ in range(total_steps):
                res = one_step(lookup_t, value_t)
            results = [res]
        else:
            ticket = iter(range(total_steps))
            lock = threading.Lock()

            def worker(stream):
                torch.cuda.set_device(local_rank)
                res = None
                with torch.cuda.stream(stream):
                    while True:
                        with lock:
                            if next(ticket, None) is None:
                                break
                        res = one_step(lookup_t, value_t)
                    stream.synchronize()
                return res

            results = [r for r in pool.map(worker, side_streams[:inflight]) if r is not None]
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()
        return time.perf_counter() - t0, results

    if args.inflight > 0:
        inflight = max(1, min(args.inflight, args.steps))
    else:
        # Up to 16 in flight, but every step in flight is a host thread that spins in the HIP runtime
        # while it waits, so leave a few of the usable cores (cgroup quota) free: a pool as large as the
        # quota gets throttled (measured on a 16-core quota: 12 in flight 17.5-18.0 Gpixels/s, 16 in
        # flight 18.7 on a quiet host but 9-11 on a loaded one).  Then balance, so that the steps split
        # into full rounds (20 steps: 2 rounds of 10).
        # Ranks of one node share the host cores.
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        cap = max(1, min(16, (usable_cores()[1] - 4) // max(local_world, 1)))
        rounds = -(-args.steps // cap)
        inflight = max(1, -(-args.steps // rounds))
    side_streams = [torch.cuda.Stream(device=device) for _ in range(inflight)]
    pool = ThreadPoolExecutor(max(inflight, 1))
    run_steps(args.warmup, 1)
    if inflight > 1:
        # several steps in flight = the library's throughput mode (four code streams per wave where a
        # kernel for it exists); the serial pass below runs in the default latency mode
        tfc.set_throughput_mode(True)
        run_steps(max(args.warmup, inflight), inflight)     # warm every stream / thread
        tfc.set_throughput_mode(False)

    # serial pass: per-kernel durations with the GPU to one launch at a time
    _lib.lib().tfc_profile_enable(1)
    serial_steps = min(args.steps, 5) if inflight > 1 else args.steps
    serial_elapsed, results = run_steps(serial_steps, 1)
    enc_ms, enc_n = profile_query("enc_kernel")
    dec_ms, dec_n = profile_query("dec_kernel")
    _lib.lib().tfc_profile_enable(0)
    # the timed region: exactly --steps steps
    if inflight > 1:
        tfc.set_throughput_mode(True)
        _lib.lib().tfc_profile_enable(1)
        elapsed, results = run_steps(args.steps, inflight)
        cenc_ms, cenc_n = profile_query("enc_kernel")
        cdec_ms, cdec_n = profile_query("dec_kernel")
        _lib.lib().tfc_profile_enable(0)
    else:
        elapsed, cenc_ms, cenc_n, cdec_ms, cdec_n = serial_elapsed, enc_ms, enc_n, dec_ms, dec_n
    pool.shutdown()
    if distributed:
        t = torch.tensor([elapsed, serial_elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, serial_elapsed = float(t[0].item()), float(t[1].item())
    blob, offsets, decoded, ok = results[-1]

    # parity gate (outside the timed region): round trip is exact, sanity flags true
    for _, _, dec_r, ok_r in results:
        assert bool(ok_r.all()), "EntropyDecodeFinalize reported a failed stream"
        assert torch.equal(dec_r.reshape(STREAMS, ELEMS), value_t), "decode(encode(x)) != x"
    total_bytes = int(offsets[-1].item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        pixels_all = world * STREAMS * PIXELS_PER_STREAM
        value_mpix = pixels_all / 1e6 / (elapsed / args.steps)
        symbols = STREAMS * ELEMS
        enc_avg = enc_ms / max(enc_n, 1)          # serial pass: a launch has the GPU to itself
        dec_avg = dec_ms / max(dec_n, 1)
        enc_tr = cenc_ms / max(cenc_n, 1)         # timed region (launches of other steps co-resident)
        dec_tr = cdec_ms / max(cdec_n, 1)
        # algorithmic bytes per launch: encode reads 4 B/symbol and
        # writes the code bytes; decode reads the code bytes and writes 4 B/symbol.
        alg_dec = 4 * symbols + total_bytes
        alg_enc = 4 * symbols + total_bytes
        dom, dom_ms, dom_bytes = ("dec_kernel", dec_tr, alg_dec) if dec_tr >= enc_tr else (
            "enc_kernel", enc_tr, alg_enc)
        achieved = dom_bytes / 1e9 / (dom_ms / 1e3) if dom_ms > 0 else 0.0
        out = {
            "metric": "Mpixels/s encode+decode round-trip (bit-exact)",
            "value": round(value_mpix, 2),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
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
                "escape_fraction": args.escape_fraction,
                "parallelism": f"batch-sharded x{world}",
                "steps_in_flight": inflight,
                "library_mode": "throughput (tfc_set_throughput_mode(1))" if inflight > 1 else "latency (default)",
            },
            "bits_per_pixel": round(8.0 * total_bytes / (STREAMS * PIXELS_PER_STREAM), 5),
            "bits_per_symbol": round(8.0 * total_bytes / symbols, 4),
            "gsymbols_per_s_roundtrip": round(world * symbols / 1e9 / (elapsed / args.steps), 3),
            "kernels_ms": {"enc_kernel": round(enc_avg, 4), "dec_kernel": round(dec_avg, 4)},
            "kernels_ms_in_flight": {"enc_kernel": round(cenc_ms / max(cenc_n, 1), 4),
                                     "dec_kernel": round(cdec_ms / max(cdec_n, 1), 4)},
            "serial": {"ms_per_step": round(1e3 * serial_elapsed / serial_steps, 4),
                       "mpixels_s": round(pixels_all / 1e6 / (serial_elapsed / serial_steps), 2),
                       "steps": serial_steps},
            "roofline": {
                "bound": "hbm", "kernel": dom, "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": pmc_traffic("dec_fast_kernel" if dom == "dec_kernel" else "enc_fast_kernel"),
                "traffic_source": "profiles/r01_o_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / "
                                  "WRITE_SIZE passes of this command; 2*FETCH + WRITE, KiB -> bytes)",
                "algorithmic_bytes": int(dom_bytes),
                "note": "serial chain per stream (512 chains): VALU-issue / synchronisation bound, not "
                        "HBM bound; per-launch duration = HIP-event average over the timed region, where "
                        "launches of the other steps in flight share the SIMDs (serial: kernels_ms); see DESIGN.md",
                "path_gbytes_s_in_flight": round((alg_enc + alg_dec) * args.steps / 1e9 / elapsed, 2),
            },
        }
        out["valu_issue_bound"] = valu_issue_floor(1e3 * elapsed / args.steps, inflight > 1)
        if world == 1:
            out["gdn_fwd"] = gdn_forward_bandwidth(device)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(lookup, value, total_bytes)
            out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 2)
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
