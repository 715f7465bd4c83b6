"""The clock the pipelined chain kernels run at: core-clock cycles / 100 MHz ticks of the first chain workgroup of an
encode and a decode launch (20 x 512 streams, BASELINE config 2), alone on the chip (TFC_PIPE_OVERLAP=0) or next to the
expansion (2).  Usage: TFC_PIPE_OVERLAP=n python tools/chain_clock_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
import bench
import compression_amd as tfc
from compression_amd import _lib

dev = torch.device("cuda", 0)
lookup = bench.build_tables(dev)
lt = torch.from_numpy(lookup)
vals = [bench.sample_symbols_device(lookup, k, dev) for k in range(20)]
for rep in range(3):
    res = bench.step_group(lt, vals, "throughput")
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 8)()
    _lib.lib().tfc_debug_pipe_clocks(out)
    e_c, e_w, d_c, d_w, waited, waits, stalled = [int(x) for x in out][:7]
    print("overlap", os.environ.get("TFC_PIPE_OVERLAP", "default"), "rep", rep,
          "enc chain: %.3f ms at %.0f MHz;" % (e_w / 1e5, 100.0 * e_c / max(e_w, 1)),
          "dec chain: %.3f ms at %.0f MHz;" % (d_w / 1e5, 100.0 * d_c / max(d_w, 1)),
          "enc chain waited %.3f ms for call words, %.3f ms for digit slots" % (waited / 2.4e6, stalled / 2.4e6), flush=True)
    del res
