import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from compression_amd.layers import conv2d_down, conv2d_up
g = torch.Generator().manual_seed(1)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
cases = [("up 5x5 x2 @48x32 n=128", conv2d_up, (128, 32, 48, 192), 5, 2),
         ("down 5x5 /2 @96x64 n=128", conv2d_down, (128, 64, 96, 192), 5, 2),
         ("down 3x3 s1 @48x32 n=128", conv2d_down, (128, 32, 48, 192), 3, 1),
         ("up 5x5 x2 @16x16 n=512", conv2d_up, (512, 16, 16, 192), 5, 2),
         ("down 5x5 /2 @32x32 n=512", conv2d_down, (512, 32, 32, 192), 5, 2),
         ("down 5x5 /2 @24x16 n=128", conv2d_down, (128, 16, 24, 192), 5, 2),
         ("up 5x5 x2 @12x8 n=128", conv2d_up, (128, 8, 12, 192), 5, 2),
         ("up 5x5 x2 @24x16 n=128", conv2d_up, (128, 16, 24, 192), 5, 2)]
for name, fn, shp, k, s in cases:
    x = torch.randn(*shp, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(k, k, 192, 192, generator=g) / (k * k * 192) ** 0.5).cuda()
    b = torch.zeros(192).cuda()
    out = []
    for gen in ("3", "4", "2"):
        os.environ["TFC_CONV_GEN"] = gen
        out.append(t(lambda: fn(x, w, b, s, None)))
    print(f"{name}: default {out[0]*1e3:.0f} us, gen3 everywhere {out[1]*1e3:.0f} us, gen2 {out[2]*1e3:.0f} us")
