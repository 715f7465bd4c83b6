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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (256, 384)
x = torch.from_numpy(synthetic.lowpass_images(8, H, W, seed=2)).to(dev).repeat((B + 7) // 8, 1, 1, 1)[:B].contiguous()
full = model.compress(x)
halves = [model.compress(x[:B // 2]), model.compress(x[B // 2:])]
seq = [np.concatenate([h[0] for h in halves]), np.concatenate([h[1] for h in halves])]
eq = [bytes(a) == bytes(b) for a, b in zip(full[0], seq[0])]
print("full batch vs two halves, sequential: y strings equal", eq.count(True), "/", B, "; z strings equal",
      [bytes(a) == bytes(b) for a, b in zip(full[1], seq[1])].count(True), "/", B, "; differing images", [i for i, e in enumerate(eq) if not e][:40])
# images repeat with period 8: the strings of image i and i + 8 must be equal within one call
print("full batch: images whose string differs from image i % 8:", [i for i in range(B) if bytes(full[0][i]) != bytes(full[0][i % 8])][:40])
print("halves:     images whose string differs from image i % 8:", [i for i in range(B) if bytes(seq[0][i]) != bytes(seq[0][i % 8])][:40])
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def run(k):
    torch.cuda.set_device(dev)
    with torch.cuda.stream(streams[k]):
        return model.compress(x[(B // 2) * k:(B // 2) * (k + 1)])
for s in streams:
    s.wait_stream(torch.cuda.current_stream())
for rep in range(3):
    with ThreadPoolExecutor(2) as pool:
        par = list(pool.map(run, range(2)))
    torch.cuda.synchronize()
    thr = [np.concatenate([h[0] for h in par]), np.concatenate([h[1] for h in par])]
    print("two halves on two threads / streams vs sequential: y equal", [bytes(a) == bytes(b) for a, b in zip(thr[0], seq[0])].count(True), "/", B, ";",
          "z equal", [bytes(a) == bytes(b) for a, b in zip(thr[1], seq[1])].count(True), "/", B)
