import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
if os.environ.get("TFC_LIB_PATH"):
    import compression_amd._lib as _L
    _L.LIB_PATH = os.environ["TFC_LIB_PATH"]
from compression_amd.layers.functional import conv2d_wgrad
g = torch.Generator().manual_seed(1)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for n, H, W in ((16, 256, 384), (32, 128, 192)):
    x = torch.randn(n, H, W, 192, generator=g).to(torch.bfloat16).cuda()
    gy = torch.randn(n, H // 2, W // 2, 192, generator=g).to(torch.bfloat16).cuda()
    ms = t(lambda: conv2d_wgrad(x, gy, (5, 5), 2, False))
    fl = 2.0 * n * (H // 2) * (W // 2) * 25 * 192 * 192
    print(f"wgrad 5x5 /2 192->192 n={n} @{W}x{H}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TFLOP/s")
# the layers of a bls2017 training step at batch 16 x 256x256 (conv2d_wgrad(a, b, kernel_support, stride, transpose): a is
# read at q s + t - k/2, b on the q grid)
for name, a_shape, b_shape, k, s in (("9x9 /4 3->192 (image side)", (16, 256, 256, 3), (16, 64, 64, 192), 9, 4),
                                    ("5x5 /2 192->192 @64x64", (16, 64, 64, 192), (16, 32, 32, 192), 5, 2),
                                    ("5x5 /2 192->192 @32x32", (16, 32, 32, 192), (16, 16, 16, 192), 5, 2),
                                    ("9x9 x4 192->3 (image side, transposed)", (16, 256, 256, 3), (16, 64, 64, 192), 9, 4)):
    a = torch.randn(*a_shape, generator=g).to(torch.bfloat16).cuda()
    b = torch.randn(*b_shape, generator=g).to(torch.bfloat16).cuda()
    ms = t(lambda: conv2d_wgrad(a, b, (k, k), s, name.endswith("transposed)")))
    print(f"wgrad {name} at 16 x 256x256: {ms:.3f} ms")
