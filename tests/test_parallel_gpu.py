"""GPU tier, world_size 2 over RCCL (backend "nccl"): the multi-GPU leg of the path — tables broadcast
once, the batch sharded, the coded strings gathered with two all-gathers — with every rank coding its
shard on its own MI355X.  Skips on a box with fewer than two devices (the CPU tier covers the same
collectives over gloo, tests/test_parallel_cpu.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        import compression_amd as tfc
        from compression_amd import parallel, synthetic
        from oracle import oracle
        lib = oracle.port()
        pmfs, _ = synthetic.gaussian_pmfs(num_tables=8, octave=2.0)
        lookup = synthetic.assemble_lookup([lib.pmf_to_quantized_cdf(p, 12) for p in pmfs], 12)
        # the tables travel from rank 0 (a rank that rebuilt them could differ in the last bit)
        table_t = torch.from_numpy(lookup).cuda() if rank == 0 else torch.zeros(len(lookup), dtype=torch.int32).cuda()
        dist.broadcast(table_t, src=0)
        value = synthetic.sample_symbols(lookup, 7, 300, seed=5, escape_fraction=0.02)
        lo, hi = parallel.shard_range(7, rank, world)
        h = tfc.create_range_encoder([hi - lo], table_t.cpu())
        h = tfc.entropy_encode_channel(h, torch.from_numpy(value[lo:hi]).cuda())
        tfc.entropy_encode_finalize(h)
        blob_all, offs_all = parallel.gather_encoded(h.blob, h.offsets)
        _, want_blob, want_offs = lib.encode(lookup, value)             # whole batch in one process
        ok = bool((offs_all.cpu().numpy() == want_offs).all() and (blob_all.cpu().numpy() == want_blob).all())
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_shard_encode_gather_world2_rccl():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two HIP devices")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, True), (1, True)]
