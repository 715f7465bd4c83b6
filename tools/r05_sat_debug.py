"""Debug aid (round 5): the saturation points of bench.py one by one — which group size decodes wrong, where."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
import bench
import compression_amd as tfc
dev = torch.device("cuda", 0)
lookup = bench.build_tables(dev)
lt = torch.from_numpy(lookup)
pts = [int(v) for v in os.environ.get("POINTS", "64,64,64").split(",")]
slots = [bench.sample_symbols_device(lookup, 7000 + k, dev) for k in range(min(max(pts), 32))]
for nb in pts:
    values = [slots[k % len(slots)] for k in range(nb)]
    for rep in range(3):
        res = bench.step_group(lt, values, "throughput")
        torch.cuda.synchronize()
        bad = []
        for k, ((h, d, dec_r, ok_r), v) in enumerate(zip(res, values)):
            eq = torch.equal(dec_r.reshape(bench.STREAMS, bench.ELEMS), v)
            if not (bool(ok_r.all()) and eq):
                diff = (dec_r.reshape(bench.STREAMS, bench.ELEMS) != v)
                rows = diff.any(1).nonzero().flatten()
                first = diff[rows[0]].nonzero().flatten()[0].item() if len(rows) else -1
                bad.append((k, int(ok_r.sum()), len(rows), int(rows[0]) if len(rows) else -1, first))
        for (k, *_rest) in bad[:2]:
            h = res[k][0]
            got = [bytes(x) for x in tfc.fetch_strings(h).reshape(-1)]
            ref_h = tfc.create_range_encoder([bench.STREAMS], lt, mode="latency")
            ref = [bytes(x) for x in tfc.entropy_encode_finalize(tfc.entropy_encode_channel(ref_h, values[k])).reshape(-1)]
            wrong = [i for i in range(bench.STREAMS) if got[i] != ref[i]]
            print("   batch", k, "strings that differ from a wave-kernel encode of the same symbols:", len(wrong), wrong[:8], flush=True)
            if wrong:
                i = wrong[0]
                n = next((j for j in range(min(len(got[i]), len(ref[i]))) if got[i][j] != ref[i][j]), -1)
                print("   stream", i, "lengths", len(got[i]), len(ref[i]), "first differing byte", n, flush=True)
        print("nb", nb, "rep", rep, "bad batches (k, ok flags, streams wrong, first stream, first element):", bad[:6], flush=True)
        del res
