#!/usr/bin/env python
"""Do CU-masked HIP streams (tfc_stream_create_cu_mask) restrict kernels to their CUs, and do two of them
run kernels concurrently?  Prints times of a bandwidth-bound and a compute-bound torch kernel chain on a
plain stream, on a 32-CU stream and on the complementary 224-CU stream, alone and together."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from compression_amd import pipeline  # noqa: E402


def chain_bw(x, n):
    for _ in range(n):
        x.add_(1.0)


def chain_alu(x, n):
    for _ in range(n):
        x = torch.sin(torch.cos(torch.sin(x)))
    return x


def timed(fn, streams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s, f in zip(streams, fn):
        with torch.cuda.stream(s):
            f()
    for s in streams:
        s.synchronize()
    return 1e3 * (time.perf_counter() - t0)


def main():
    part = pipeline.CoderPartition(coder_cus=int(sys.argv[1]) if len(sys.argv) > 1 else 32, depth=1)
    lane = part.lane(0)
    plain = torch.cuda.Stream()
    plain2 = torch.cuda.Stream()
    big = torch.zeros(1 << 28, device="cuda")           # 1 GiB fp32
    big2 = torch.zeros(1 << 28, device="cuda")
    small = torch.zeros(1 << 22, device="cuda")
    small2 = torch.zeros(1 << 22, device="cuda")
    for s in (plain, lane.coder, lane.transform):
        with torch.cuda.stream(s):
            chain_bw(big, 2)
            chain_alu(small, 2)
    torch.cuda.synchronize()
    print(f"CUs: {part.total_cus}, coder share {part.coder_cus}")
    for name, s in (("plain", plain), ("coder-masked", lane.coder), ("transform-masked", lane.transform)):
        print(f"{name:18s} bandwidth chain {timed([lambda: chain_bw(big, 20)], [s]):8.2f} ms   "
              f"alu chain (4M elements x 300) {timed([lambda: chain_alu(small, 100)], [s]):8.2f} ms")
    a = timed([lambda: chain_bw(big, 20)], [lane.transform])
    b = timed([lambda: chain_alu(small, 100)], [lane.coder])
    both = timed([lambda: chain_bw(big, 20), lambda: chain_alu(small, 100)], [lane.transform, lane.coder])
    print(f"masked streams: A alone {a:.2f}, B alone {b:.2f}, together {both:.2f} ms (sum {a + b:.2f})")
    a = timed([lambda: chain_bw(big, 20)], [plain])
    b = timed([lambda: chain_alu(small, 100)], [plain2])
    both = timed([lambda: chain_bw(big, 20), lambda: chain_alu(small, 100)], [plain, plain2])
    print(f"plain streams:  A alone {a:.2f}, B alone {b:.2f}, together {both:.2f} ms (sum {a + b:.2f})")
    # two big kernels chains on the two masked streams
    a = timed([lambda: chain_bw(big, 20)], [lane.transform])
    b = timed([lambda: chain_bw(big2, 20)], [lane.coder])
    both = timed([lambda: chain_bw(big, 20), lambda: chain_bw(big2, 20)], [lane.transform, lane.coder])
    print(f"masked, both bandwidth chains: A {a:.2f}, B {b:.2f}, together {both:.2f} ms")
    part.close()


if __name__ == "__main__":
    main()


def lanes_probe():
    """Two lanes, each: transform chain -> coder chain -> transform chain, enqueued lane after lane by one
    thread (what bench.run_model_steps does).  With overlap the total tends to 2 T + C + T; serialised it is
    2 (2 T + C)."""
    x = [torch.zeros(1 << 24, device="cuda") for _ in range(4)]
    for masked in (True, False):
        if masked:
            part = pipeline.CoderPartition(coder_cus=32, depth=2)
            lanes = part.lanes
        else:
            lanes = [pipeline.Lane(torch.cuda.Stream(), torch.cuda.Stream()) for _ in range(2)]
        side = torch.cuda.Stream()

        def step(lane, k):
            with lane.on("transform"):
                chain_alu(x[2 * k], 40)
            with lane.on("coder"):
                chain_alu(x[2 * k + 1][:1 << 21], 60)
            with lane.on("transform"):
                chain_alu(x[2 * k], 40)

        with torch.cuda.stream(side):
            for reps in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                step(lanes[0].begin(side), 0)
                torch.cuda.synchronize()
                one = 1e3 * (time.perf_counter() - t0)
                t0 = time.perf_counter()
                for k in range(2):
                    step(lanes[k].begin(side), k)
                t_enq = 1e3 * (time.perf_counter() - t0)
                torch.cuda.synchronize()
                two = 1e3 * (time.perf_counter() - t0)
            # the parts alone
            t0 = time.perf_counter()
            with torch.cuda.stream(lanes[0].transform):
                chain_alu(x[0], 40)
            torch.cuda.synchronize()
            tt = 1e3 * (time.perf_counter() - t0)
            t0 = time.perf_counter()
            with torch.cuda.stream(lanes[0].coder):
                chain_alu(x[1][:1 << 21], 60)
            torch.cuda.synchronize()
            tc = 1e3 * (time.perf_counter() - t0)
        print(f"{'masked' if masked else 'plain '} lanes: T {tt:.2f} C {tc:.2f} | one step {one:.2f} ms, two steps in flight "
              f"{two:.2f} ms (enqueue {t_enq:.2f}); ideal overlapped {2 * tt + max(tc, tt) + tt:.2f}, serial {2 * (2 * tt + tc):.2f}")
        if masked:
            part.close()


if __name__ == "__main__":
    lanes_probe()
