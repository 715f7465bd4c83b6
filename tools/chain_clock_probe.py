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
vals = [bench.sample_symbols_device(lookup, k, dev) for k in range(int(os.environ.get("BATCHES", "20")))]
if os.environ.get("IDENTICAL_STREAMS"):
    # every stream of a batch the same symbols: the 64 lanes of a chain wave read the same LDS addresses (broadcasts,
    # no bank conflicts) — what the random rows' conflicts cost a step
    vals = [v[:1].expand_as(v).contiguous() for v in vals]
for rep in range(3):
    res = bench.step_group(lt, vals, "throughput")
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 8)()
    _lib.lib().tfc_debug_pipe_clocks(out)
    e_c, e_w, d_c, d_w, waited, waits, stalled = [int(x) for x in out][:7]
    print("overlap", os.environ.get("TFC_PIPE_OVERLAP", "default"), "rep", rep,
          "enc chain: %.3f ms at %.0f MHz;" % (e_w / 1e5, 100.0 * e_c / max(e_w, 1)),
          "dec chain: %.3f ms at %.0f MHz;" % (d_w / 1e5, 100.0 * d_c / max(d_w, 1)),
          "enc chain waited %.3f ms for call words, %.3f ms for digit slots (0: not a TFC_PIPE_TIMING build)" % (waited / 2.4e6, stalled / 2.4e6), flush=True)
    eo = (C.c_ulonglong * 4)()
    C.CDLL(_lib.LIB_PATH).tfc_debug_enc_clocks(eo)
    if eo[1]:
        print("   enc chain (TFC_PIPE_TIMING): %d blocks, %.1f cycles per row inside them, %d repeated call by call (%.0f cycles each), %.1f per row over the kernel"
              % (eo[1], eo[0] / (16.0 * eo[1]), eo[2], eo[3] / max(int(eo[2]), 1), e_c / (16.0 * eo[1])), flush=True)
    if out[5]:
        blocks, rows, t_rest = int(out[7]) & 0xFFFF, (int(out[7]) >> 16) & 0xFFFF, int(out[7]) >> 32
        t_asm, t_commit = int(out[5]) & 0xFFFFFFFF, int(out[5]) >> 32
        print("   dec chain (build with -DTFC_PIPE_TIMING=1): %d rows, %d hand-scheduled blocks, %.1f cycles per row inside them, %.1f per row over the kernel; "
              "memory phases: %.0f cycles per block waiting for + parking the windows, %.0f flush + requests"
              % (rows, blocks, t_asm / max(16 * blocks, 1), d_c / max(rows, 1), t_commit / max(blocks, 1), t_rest / max(blocks, 1)), flush=True)
        print("   %d blocks more than rows / 16 (repeated step by step)" % (blocks - rows // 16), flush=True)
    del res
