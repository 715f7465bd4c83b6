"""CPU tier, world_size 2 over gloo: batch sharding and the variable-length gather of
coded strings used by the multi-GPU path (bench.py --gpus N)."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from compression_amd import parallel


def test_shard_range_covers_everything():
    for total in (0, 1, 7, 512, 1000):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle
        from compression_amd import synthetic
        lib = oracle.port()
        pmfs, _ = synthetic.gaussian_pmfs(num_tables=8, octave=2.0)
        lookup = synthetic.assemble_lookup([lib.pmf_to_quantized_cdf(p, 12) for p in pmfs], 12)
        value = synthetic.sample_symbols(lookup, 7, 300, seed=5, escape_fraction=0.02)
        lo, hi = parallel.shard_range(7, rank, world)
        strings, blob, offs = lib.encode(lookup, value[lo:hi])          # this rank's shard (CPU oracle)
        blob_all, offs_all = parallel.gather_encoded(torch.from_numpy(np.ascontiguousarray(blob)),
                                                     torch.from_numpy(offs))
        _, want_blob, want_offs = lib.encode(lookup, value)             # whole batch in one process
        ok = bool((offs_all.numpy() == want_offs).all() and (blob_all.numpy() == want_blob).all())
        w = torch.nn.Linear(3, 2)
        parallel.broadcast_tables(w)
        t = w.weight.detach().clone()
        dist.all_reduce(t)
        ok = ok and torch.allclose(t, world * w.weight.detach())
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_gather_encoded_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, True), (1, True)]


class _HostReads:
    """Counts the ways a tensor's VALUE reaches the host (each one is a device synchronisation on a GPU)."""
    NAMES = ("item", "tolist", "__int__", "__bool__", "__float__", "__index__", "cpu", "numpy")

    def __enter__(self):
        self.count, self._saved = 0, {}
        for name in self.NAMES:
            orig = getattr(torch.Tensor, name)
            self._saved[name] = orig

            def spy(t, *a, _orig=orig, **k):
                self.count += 1
                return _orig(t, *a, **k)
            setattr(torch.Tensor, name, spy)
        return self

    def __exit__(self, *exc):
        for name, orig in self._saved.items():
            setattr(torch.Tensor, name, orig)


def _async_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle
        from compression_amd import synthetic
        lib = oracle.port()
        pmfs, _ = synthetic.gaussian_pmfs(num_tables=8, octave=2.0)
        lookup = synthetic.assemble_lookup([lib.pmf_to_quantized_cdf(p, 12) for p in pmfs], 12)
        steps, gathers, ok = 3, 0, True
        for step in range(steps):
            value = synthetic.sample_symbols(lookup, 7, 300, seed=11 + step, escape_fraction=0.02)
            lo, hi = parallel.shard_range(7, rank, world)
            _, blob, offs = lib.encode(lookup, value[lo:hi])                # this rank's shard (CPU oracle)
            _, want_blob, want_offs = lib.encode(lookup, value)             # whole batch in one process
            # a slab at capacity, as a finalized handle's device_strings view has it: junk behind the total
            slab = torch.full((4096,), 0xAB, dtype=torch.uint8)
            slab[:len(blob)] = torch.from_numpy(np.ascontiguousarray(blob))
            with _HostReads() as reads:
                got = parallel.gather_encoded_async(slab, torch.from_numpy(offs), capacity_bytes=2048, capacity_streams=4)
            gathers += 1
            ok = ok and reads.count == 0                                    # nothing in the gather looks at a value
            with _HostReads() as reads:
                got.wait()                                                  # ... nor does ordering a consumer behind it
            ok = ok and reads.count == 0
            blob_all, offs_all = got.packed()
            ok = ok and bool((offs_all.numpy() == want_offs).all() and (blob_all.numpy() == want_blob).all())
            ok = ok and not bool(got.overflow)
            # every stream through (starts, lengths) of the padded layout
            starts = got.starts
            k = 0
            for r in range(world):
                for i in range(int(got.counts[r])):
                    a, n = int(starts[r, i]), int(got.lengths[r, i])
                    ok = ok and bytes(got.blob[r, a:a + n].numpy()) == bytes(want_blob[want_offs[k]:want_offs[k + 1]])
                    k += 1
            # a slot that is too small: flagged on every rank, packed() gathers again, exactly
            small = parallel.gather_encoded_async(slab, torch.from_numpy(offs), capacity_bytes=16, capacity_streams=4)
            small.wait()
            ok = ok and bool(small.overflow)
            blob_all, offs_all = small.packed()
            ok = ok and bool((offs_all.numpy() == want_offs).all() and (blob_all.numpy() == want_blob).all())
        q.put((rank, ok and gathers == steps))
    finally:
        dist.destroy_process_group()


def test_gather_encoded_async_world2_gloo():
    """A step's gather enqueues two collectives of host-known sizes and reads nothing back: gathers == steps, zero host
    reads inside the gather, packed() == the whole batch coded by one process."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_async_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, True), (1, True)]


def _run_bench(*argv, timeout=300):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_bench_gpus_n_starts_n_ranks():
    """`python bench.py --gpus 2` with no launcher around it must START two ranks (torch.distributed.run on
    127.0.0.1) — here over gloo, without devices (`--spawn-check`): both ranks rendezvous and count each other."""
    import json
    r = _run_bench("--gpus", "2", "--spawn-check")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line == {**line, "spawn_check": True, "world": 2, "ranks_seen": 2}


def test_bench_gpus_n_fails_loudly_without_n_devices():
    """A measurement asked for on more devices than the node has is an error, not a one-GPU number."""
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this node has two devices")
    r = _run_bench("--gpus", "2", "--no-extras", timeout=120)
    assert r.returncode != 0
    assert "--gpus 2 asked for" in r.stderr


def test_bench_rejects_a_world_that_is_not_gpus():
    """Under a launcher, WORLD_SIZE must be what --gpus says."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--spawn-check"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
