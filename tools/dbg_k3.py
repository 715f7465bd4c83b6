import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import compression_amd as tfc
g = np.load("tests/golden/kat_raw.npz")
for k in ("K3", "K4", "K6"):
    prec = int(g[k + "_precision"][0])
    lookup = np.concatenate([[prec], g[k + "_cdf"]]).astype(np.int32)[None, :]
    syms = g[k + "_syms"][None, :]
    arr = np.empty(1, dtype=object); arr[0] = g[k + "_bytes"].tobytes()
    h = tfc.create_range_decoder(arr, torch.as_tensor(lookup))
    h, out = tfc.entropy_decode_channel(h, [syms.shape[1]], torch.int32)
    print(k, "want", syms[0][:24].tolist()); print(k, "got ", out.cpu().numpy()[0][:24].tolist())
