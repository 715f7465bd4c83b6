"""Phases of the expansion kernel (a library built with clock reads at its barriers — see profiles/r05_notes.md; the
shipped library prints zeros): average cycles per workgroup from entry to the barrier behind phase A (symbols ->
table entries), through phase B (escape codes in order, row counts, look-back), through phase C (call words out)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import compression_amd as tfc
from compression_amd import _lib

dev = torch.device("cuda", 0)
lookup = bench.build_tables(dev)
lt = torch.from_numpy(lookup)
vals = [bench.sample_symbols_device(lookup, k, dev) for k in range(int(os.environ.get("BATCHES", "20")))]
lib = C.CDLL(_lib.LIB_PATH)
prev = [0, 0, 0, 0]
for rep in range(3):
    res = bench.step_group(lt, vals, "throughput")
    torch.cuda.synchronize()
    eo = (C.c_ulonglong * 4)()
    lib.tfc_debug_enc_clocks(eo)
    cur = [int(x) for x in eo]
    d = [c - p for c, p in zip(cur, prev)]
    prev = cur
    if d[3]:
        print("overlap", os.environ.get("TFC_PIPE_OVERLAP", "default"), "rep", rep, "workgroups", d[3],
              "cycles per workgroup: phase A %.0f, phase B %.0f, phase C %.0f" % (d[0] / d[3], d[1] / d[3], d[2] / d[3]), flush=True)
    del res
