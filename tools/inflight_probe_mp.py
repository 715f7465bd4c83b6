#!/usr/bin/env python
"""Same question as inflight_probe.py with PROCESSES instead of threads: is the in-flight plateau
a host-side (GIL / runtime lock) limit or a GPU one?  python tools/inflight_probe_mp.py"""
import os, sys, time
import multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, nproc, threads, steps, barrier, out):
    import torch
    from concurrent.futures import ThreadPoolExecutor
    import bench
    from compression_amd import synthetic
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    lookup = bench.build_tables(dev)
    value = synthetic.sample_symbols(lookup, bench.STREAMS, bench.ELEMS, seed=0)
    lt, vt = torch.from_numpy(lookup), torch.from_numpy(value).to(dev)
    for _ in range(2):
        bench.one_step(lt, vt)
    streams = [torch.cuda.Stream() for _ in range(threads)]

    def run(s, n):
        torch.cuda.set_device(0)
        with torch.cuda.stream(s):
            for _ in range(n):
                bench.one_step(lt, vt)
            s.synchronize()

    with ThreadPoolExecutor(threads) as pool:
        list(pool.map(lambda s: run(s, 1), streams))
        torch.cuda.synchronize()
        barrier.wait()
        t0 = time.perf_counter()
        list(pool.map(lambda s: run(s, steps), streams))
        torch.cuda.synchronize()
        out.put((t0, time.perf_counter()))


if __name__ == "__main__":
    mp.set_start_method("spawn")
    import bench
    for nproc, threads in ((1, 4), (2, 2), (4, 1), (4, 2), (2, 4)):
        steps = 6
        barrier = mp.Barrier(nproc)
        out = mp.Queue()
        ps = [mp.Process(target=worker, args=(r, nproc, threads, steps, barrier, out)) for r in range(nproc)]
        for p in ps:
            p.start()
        res = [out.get() for _ in ps]
        for p in ps:
            p.join()
        t0, t1 = min(r[0] for r in res), max(r[1] for r in res)
        total = nproc * threads * steps
        print(f"{nproc} proc x {threads} threads: {total} steps in {(t1 - t0) * 1e3:7.1f} ms -> "
              f"{(t1 - t0) * 1e3 / total:5.2f} ms/step, {total * bench.STREAMS * bench.PIXELS_PER_STREAM / 1e6 / (t1 - t0):8.0f} Mpixels/s")
