#!/usr/bin/env python
"""One C2 encode + decode with the kernels built with -DTFC_PHASE_TIMING (see tools/rebuild_coder_with_flags.sh):
stream 0 prints its cycle split per phase."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from compression_amd import synthetic
dev = torch.device("cuda", 0)
lookup = bench.build_tables(dev)
value = synthetic.sample_symbols(lookup, bench.STREAMS, bench.ELEMS, seed=0)
bench.one_step(torch.from_numpy(lookup), torch.from_numpy(value).to(dev))
torch.cuda.synchronize()
