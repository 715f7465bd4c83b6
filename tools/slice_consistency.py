"""Are the strings of an image independent of the batch it is coded in, and of other host threads coding
other slices at the same time?  bmshj2018, 16 images of 256x384, bf16."""
import os, sys
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import compression_amd as tfc
from compression_amd import synthetic

torch.manual_seed(0)
dev = torch.device("cuda", 0)
model = tfc.models.BMSHJ2018Model(num_filters=192, compute_dtype=torch.bfloat16).to(dev).init_compression()
x = torch.from_numpy(synthetic.lowpass_images(16, 256, 384, seed=2)).to(dev)
full = model.compress(x)
halves = [model.compress(x[:8]), model.compress(x[8:])]
seq = [np.concatenate([h[0] for h in halves]), np.concatenate([h[1] for h in halves])]
print("full batch vs two halves, sequential: y strings equal", [bytes(a) == bytes(b) for a, b in zip(full[0], seq[0])].count(True), "/ 16;",
      "z strings equal", [bytes(a) == bytes(b) for a, b in zip(full[1], seq[1])].count(True), "/ 16")
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def run(k):
    torch.cuda.set_device(dev)
    with torch.cuda.stream(streams[k]):
        return model.compress(x[8 * k:8 * k + 8])
for s in streams:
    s.wait_stream(torch.cuda.current_stream())
for rep in range(3):
    with ThreadPoolExecutor(2) as pool:
        par = list(pool.map(run, range(2)))
    torch.cuda.synchronize()
    thr = [np.concatenate([h[0] for h in par]), np.concatenate([h[1] for h in par])]
    print("two halves on two threads / streams vs sequential: y equal", [bytes(a) == bytes(b) for a, b in zip(thr[0], seq[0])].count(True), "/ 16;",
          "z equal", [bytes(a) == bytes(b) for a, b in zip(thr[1], seq[1])].count(True), "/ 16")
