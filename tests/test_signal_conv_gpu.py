"""GPU tier: SignalConv2D kernels (2-D, same_zeros, explicit-padding paths,
python/layers/signal_conv.py:663-690, 778-847) against a torch fp32 evaluation of
the same definition (conv2d with explicit padding / conv_transpose2d), which plays
the role of the reference test's NumPy/SciPy oracle (signal_conv_test.py:171-264),
plus the identity-kernel alignment checks of signal_conv_test.py:266-314."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def ref_down(x, k, bias, stride, relu):
    kh, kw = k.shape[:2]
    xp = F.pad(x.permute(0, 3, 1, 2), (kw // 2, (kw - 1) // 2, kh // 2, (kh - 1) // 2))
    y = F.conv2d(xp, k.permute(3, 2, 0, 1), bias, stride=stride)
    y = y.permute(0, 2, 3, 1)
    return F.relu(y) if relu else y


def ref_up(x, k, bias, stride, relu):
    # y[j] = sum_i x[i] w[j - i*s + k//2], length in*s (extra_pad_end=True)
    kh, kw = k.shape[:2]
    n, h, w, _ = x.shape
    full = F.conv_transpose2d(x.permute(0, 3, 1, 2), k.permute(2, 3, 0, 1), bias, stride=stride)
    full = F.pad(full, (0, kw, 0, kh))       # room for the crop below
    y = full[:, :, kh // 2:kh // 2 + h * stride, kw // 2:kw // 2 + w * stride]
    y = y.permute(0, 2, 3, 1)
    return F.relu(y) if relu else y


CASES = [
    # (N, H, W, Cin, Cout, k, stride)
    (2, 19, 23, 3, 32, 9, 4),      # bls2017 analysis layer 0 shape class (image input)
    (1, 16, 18, 3, 192, 5, 2),     # bmshj2018 analysis layer 0
    (2, 17, 13, 32, 64, 5, 2),
    (1, 9, 11, 192, 192, 5, 2),    # the C -> C layers
    (1, 8, 8, 48, 32, 3, 1),       # hyper-analysis 3x3 s1
    (1, 10, 7, 16, 40, 4, 2),      # even kernel, asymmetric padding
    (1, 6, 5, 64, 320, 5, 2),      # more than one column group
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("relu", [False, True])
def test_down_f32(case, relu):
    from compression_amd.layers import conv2d_down
    n, h, w, cin, cout, k, s = case
    torch.manual_seed(0)
    x = torch.randn(n, h, w, cin)
    ker = torch.randn(k, k, cin, cout) / np.sqrt(k * k * cin)
    bias = torch.randn(cout)
    y = conv2d_down(x.cuda(), ker, bias, s, "relu" if relu else None).cpu()
    want = ref_down(x, ker, bias, s, relu)
    assert y.shape == want.shape == (n, -(-h // s), -(-w // s), cout)
    assert (y - want).abs().max() <= 2e-5 * max(1.0, want.abs().max())


UP_CASES = [
    (2, 7, 9, 32, 32, 5, 2),
    (1, 6, 5, 192, 192, 5, 2),     # synthesis C -> C
    (1, 5, 6, 192, 3, 9, 4),       # bls2017 synthesis last layer (C -> 3, x4)
    (1, 8, 7, 64, 3, 5, 2),        # bmshj2018 synthesis last layer
    (1, 9, 8, 48, 48, 3, 1),       # corr=False without upsampling (flipped kernel)
    (1, 5, 5, 16, 24, 4, 2),       # even kernel
    (1, 4, 6, 32, 96, 5, 3),
]


@pytest.mark.parametrize("case", UP_CASES)
def test_up_f32(case):
    from compression_amd.layers import conv2d_up
    n, h, w, cin, cout, k, s = case
    torch.manual_seed(1)
    x = torch.randn(n, h, w, cin)
    ker = torch.randn(k, k, cin, cout) / np.sqrt(k * k * cin)
    bias = torch.randn(cout)
    y = conv2d_up(x.cuda(), ker, bias, s).cpu()
    want = ref_up(x, ker, bias, s, False)
    assert y.shape == want.shape == (n, h * s, w * s, cout)
    assert (y - want).abs().max() <= 2e-5 * max(1.0, want.abs().max())


GEN3_CASES = [  # up, n, h, w, cin, cout, k, s  — shapes the third-generation bf16 kernel takes (maps >= 24 wide)
    (False, 2, 64, 64, 192, 192, 5, 2),      # analysis 5x5 /2 (TFC_CONV_GEN=4 below: built, not the default there)
    (False, 1, 62, 186, 192, 192, 5, 2),     # blocks hanging over the right and lower edge
    (False, 2, 32, 48, 192, 192, 3, 1),      # hyper-analysis 3x3
    (False, 1, 64, 64, 128, 128, 5, 2),      # bls2017 width
    (True, 2, 32, 48, 192, 192, 5, 2),       # synthesis 5x5 x2: four phase groups of 9 / 6 / 6 / 4 taps
    (True, 1, 31, 93, 192, 192, 5, 2),
    (True, 1, 32, 32, 128, 128, 5, 2),
    (True, 1, 24, 40, 192, 192, 3, 1),
    (False, 1, 48, 64, 64, 192, 5, 2),       # 4 channel blocks only
]


@pytest.mark.parametrize("case", GEN3_CASES)
def test_bf16_third_generation(case, monkeypatch):
    """conv3_bf16_kernel (input patch staged in LDS per 16-channel block, K = channel block x taps) against the
    float32 definition, and against the second-generation kernel on the same inputs (their K order differs: equal
    up to the rounding of the bf16 result)."""
    from compression_amd.layers import conv2d_down, conv2d_up
    up, n, h, w, cin, cout, k, s = case
    torch.manual_seed(5)
    x = torch.randn(n, h, w, cin).bfloat16()
    ker = (torch.randn(k, k, cin, cout) / np.sqrt(k * k * cin)).bfloat16().float()
    bias = torch.randn(cout)
    fn, ref = (conv2d_up, ref_up) if up else (conv2d_down, ref_down)
    want = ref(x.float(), ker, bias, s, True)
    got = {}
    for gen in ("4", "2"):
        monkeypatch.setenv("TFC_CONV_GEN", gen)
        got[gen] = fn(x.cuda(), ker, bias, s, "relu").float().cpu()
        assert got[gen].shape == want.shape
        assert (got[gen] - want).abs().max() <= 2 ** -7 * max(1.0, want.abs().max())
    assert (got["4"] - got["2"]).abs().max() <= 2 ** -6 * max(1.0, want.abs().max())
    # and image by image the same whatever the batch around it
    monkeypatch.setenv("TFC_CONV_GEN", "4")
    one = fn(x[:1].cuda(), ker, bias, s, "relu").float().cpu()
    assert torch.equal(one, got["4"][:1])


def test_bf16_paths():
    from compression_amd.layers import conv2d_down, conv2d_up
    torch.manual_seed(2)
    x = torch.randn(2, 12, 10, 64).bfloat16()
    ker = (torch.randn(5, 5, 64, 96) / 40).bfloat16().float()     # weights exactly representable
    bias = torch.randn(96)
    want = ref_down(x.float(), ker, bias, 2, True)
    y = conv2d_down(x.cuda(), ker, bias, 2, "relu").float().cpu()
    assert (y - want).abs().max() <= 2 ** -7 * max(1.0, want.abs().max())
    want = ref_up(x.float(), ker, bias, 2, False)
    y = conv2d_up(x.cuda(), ker, bias, 2).float().cpu()
    assert (y - want).abs().max() <= 2 ** -7 * max(1.0, want.abs().max())
    xi = torch.rand(1, 20, 20, 3).bfloat16()
    k0 = (torch.randn(9, 9, 3, 32) / 16).bfloat16().float()
    y = conv2d_down(xi.cuda(), k0, None, 4).float().cpu()
    want = ref_down(xi.float(), k0, None, 4, False)
    assert (y - want).abs().max() <= 2 ** -7 * max(1.0, want.abs().max())


@pytest.mark.parametrize("k,s,cin,cout,hw", [(5, 2, 192, 3, (13, 9)), (9, 4, 64, 3, (7, 10)), (5, 2, 32, 1, (6, 6)),
                                             (4, 2, 48, 4, (5, 8)), (5, 3, 32, 2, (4, 7))])
def test_up_into_few_channels_bf16(k, s, cin, cout, hw):
    """The last synthesis layer (C -> 3 and the like, bf16; kernels up to 5x5) runs as one 1x1 product per input
    pixel plus a gather over the taps that land on an output pixel (signal_conv.hip, conv_up_small_cout), larger
    kernels as the implicit GEMM over output pixels: same result as the definition (signal_conv.py:778-847),
    with bias and with ReLU."""
    from compression_amd.layers import conv2d_up
    torch.manual_seed(k * 10 + s)
    x = torch.randn(3, hw[0], hw[1], cin).bfloat16()
    ker = (torch.randn(k, k, cin, cout) / (k * cin ** 0.5)).bfloat16().float()
    bias = torch.randn(cout)
    for act in (None, "relu"):
        want = ref_up(x.float(), ker, bias, s, act == "relu")
        y = conv2d_up(x.cuda(), ker, bias, s, act).float().cpu()
        assert y.shape == want.shape
        assert (y - want).abs().max() <= 2 ** -7 * max(1.0, want.abs().max())


@pytest.mark.parametrize("case", [(5, 2, 192, 3, (40, 70)), (5, 2, 64, 3, (9, 33)), (3, 2, 32, 4, (17, 50)),
                                  (5, 2, 192, 2, (8, 32)), (4, 2, 64, 3, (11, 37)),
                                  # more product columns than an LDS row: a pass per kernel-row residue
                                  (9, 4, 128, 3, (20, 40)), (9, 4, 192, 3, (9, 33)), (7, 3, 64, 4, (9, 34))])
def test_up_into_few_channels_fused_equals_unfused(case, monkeypatch):
    """conv_up_fused_kernel (tap products of a block kept in LDS; kernels with more product columns than an LDS row
    take the implicit GEMM) against the float32 definition, at the tolerance of one bf16 rounding — blocks hanging
    over every edge, several column tiles, even kernels."""
    from compression_amd.layers import conv2d_up
    k, s, cin, cout, hw = case
    torch.manual_seed(k + cin)
    x = torch.randn(2, hw[0], hw[1], cin).bfloat16()
    ker = (torch.randn(k, k, cin, cout) / (k * cin ** 0.5)).bfloat16().float()
    bias = torch.randn(cout)
    want = ref_up(x.float(), ker, bias, s, False)
    y = conv2d_up(x.cuda(), ker, bias, s).float().cpu()
    assert y.shape == want.shape
    assert (y - want).abs().max() <= 2 ** -8 * max(1.0, want.abs().max())
    one = conv2d_up(x[1:].cuda(), ker, bias, s).float().cpu()
    assert torch.equal(one, y[1:])


@pytest.mark.parametrize("case", [(5, 2, 3, 192, (64, 96)), (5, 2, 3, 192, (37, 71)), (9, 4, 3, 128, (128, 128)),
                                  (9, 4, 3, 128, (101, 190)), (5, 2, 1, 128, (48, 64)), (5, 2, 4, 192, (33, 50))])
def test_image_side_analysis_layer_bf16(case):
    """conv_image_kernel (image patch in LDS as interleaved (x, c) rows, a kernel row = ceil(kw * Cin / 16) K steps)
    against the float32 definition — blocks over every edge, 1 / 3 / 4 input channels, both strides — and image by
    image the same whatever the batch."""
    from compression_amd.layers import conv2d_down
    k, s, cin, cout, hw = case
    torch.manual_seed(k + cout)
    x = torch.rand(3, hw[0], hw[1], cin).bfloat16()
    ker = (torch.randn(k, k, cin, cout) / (k * cin ** 0.5)).bfloat16().float()
    bias = torch.randn(cout)
    for act in (None, "relu"):
        want = ref_down(x.float(), ker, bias, s, act == "relu")
        y = conv2d_down(x.cuda(), ker, bias, s, act).float().cpu()
        assert y.shape == want.shape
        assert (y - want).abs().max() <= 2 ** -7 * max(1.0, want.abs().max())
    one = conv2d_down(x[2:].cuda(), ker, bias, s, "relu").float().cpu()
    assert torch.equal(one, y[2:])


def test_identity_kernel_alignment():
    """signal_conv_test.py:266-314: with a centred delta kernel, `same_zeros` output
    sample 0 is aligned with input sample 0 (down: subsampling; up: zero-stuffing)."""
    from compression_amd.layers import conv2d_down, conv2d_up
    C = 16
    x = torch.arange(1.0, 1 + 2 * 9 * 11 * C).reshape(2, 9, 11, C) / 100
    for k in (3, 4, 5):
        ker = torch.zeros(k, k, C, C)
        ker[k // 2, k // 2] = torch.eye(C)
        for s in (1, 2, 3):
            y = conv2d_down(x.cuda(), ker, None, s).cpu()
            assert torch.equal(y, x[:, ::s, ::s])
            yu = conv2d_up(x.cuda(), ker, None, s).cpu()
            want = torch.zeros(2, 9 * s, 11 * s, C)
            want[:, ::s, ::s] = x
            assert torch.equal(yu, want), (k, s)


def test_argument_errors():
    from compression_amd.layers import conv2d_down
    with pytest.raises(ValueError, match="multiple of 16"):
        conv2d_down(torch.zeros(1, 4, 4, 10).cuda(), torch.zeros(3, 3, 10, 8))
    with pytest.raises(ValueError, match="rank 4"):
        conv2d_down(torch.zeros(4, 4, 16).cuda(), torch.zeros(3, 3, 16, 8))


GRAD_CASES = [
    # (up, N, H, W, Cin, Cout, k, stride, relu)
    (False, 2, 17, 13, 32, 64, 5, 2, False),
    (False, 1, 9, 11, 192, 192, 5, 2, False),     # C -> C analysis
    (False, 2, 19, 23, 3, 64, 9, 4, False),       # image layer: narrow A side
    (False, 1, 8, 8, 64, 32, 3, 1, True),         # hyper 3x3 s1 with fused ReLU
    (False, 1, 10, 7, 32, 64, 4, 2, False),       # even kernel
    (True, 2, 7, 9, 32, 32, 5, 2, False),
    (True, 1, 6, 5, 192, 192, 5, 2, False),       # C -> C synthesis
    (True, 1, 5, 6, 192, 3, 9, 4, False),         # last synthesis layer: narrow Cout
    (True, 1, 9, 8, 64, 64, 3, 1, True),
    (True, 1, 6, 7, 352, 224, 5, 1, True),        # ms2020 slice transform widths: weight gradient in channel blocks
    (False, 1, 8, 6, 320, 256, 5, 2, False),      # ms2020 hyper-analysis layer 1
]


@pytest.mark.parametrize("case", GRAD_CASES)
def test_conv_gradients_f32(case):
    """dx, dw, dbias of both directions against torch autograd of the reference definition."""
    from compression_amd.layers import conv2d_down, conv2d_up
    up, n, h, w, cin, cout, k, s, relu = case
    torch.manual_seed(3)
    x0 = torch.randn(n, h, w, cin)
    k0 = torch.randn(k, k, cin, cout) / np.sqrt(k * k * cin)
    b0 = torch.randn(cout)
    res = []
    for dev, fn in (("cpu", ref_up if up else ref_down), ("cuda", None)):
        x = x0.detach().clone().to(dev).requires_grad_(True)
        ker = k0.detach().clone().to(dev).requires_grad_(True)
        bias = b0.detach().clone().to(dev).requires_grad_(True)
        if fn is not None:
            y = fn(x, ker, bias, s, relu)
        else:
            y = (conv2d_up if up else conv2d_down)(x, ker, bias, s, "relu" if relu else None)
        torch.manual_seed(4)
        gy = torch.randn(y.shape)
        (y * gy.to(dev)).sum().backward()
        res.append((x.grad.cpu(), ker.grad.cpu(), bias.grad.cpu()))
    for want, got in zip(*res):
        assert got.shape == want.shape
        assert (got - want).abs().max() <= 5e-5 * max(1.0, want.abs().max())


def test_conv_gradients_bf16_and_module():
    from compression_amd.layers import SignalConv2D, conv2d_down
    torch.manual_seed(5)
    x0 = torch.randn(2, 12, 10, 64).bfloat16()
    k0 = (torch.randn(5, 5, 64, 128) / 40).bfloat16().float()
    x = x0.cuda().requires_grad_(True)
    ker = k0.cuda().requires_grad_(True)
    y = conv2d_down(x, ker, None, 2)
    gy = torch.randn(y.shape).bfloat16()
    (y * gy.cuda()).sum().backward()
    xr = x0.float().requires_grad_(True)
    kr = k0.clone().requires_grad_(True)
    (ref_down(xr, kr, None, 2, False) * gy.float()).sum().backward()
    assert (x.grad.float().cpu() - xr.grad).abs().max() <= 2 ** -6 * xr.grad.abs().max()
    assert (ker.grad.cpu() - kr.grad).abs().max() <= 2 ** -6 * kr.grad.abs().max()
    # module: gradients reach the rdft-parameterised kernel and the bias
    layer = SignalConv2D(32, (5, 5), corr=False, strides_up=2, padding="same_zeros", use_bias=True,
                         in_channels=64).cuda()
    out = layer(torch.randn(1, 6, 6, 64, device="cuda", requires_grad=True))
    out.square().sum().backward()
    grads = [p.grad for p in layer.parameters()]
    assert grads and all(g is not None and torch.isfinite(g).all() and g.abs().sum() > 0 for g in grads)


@pytest.mark.parametrize("ca,cb,k,s,transpose", [
    (32, 64, 5, 2, False), (192, 192, 5, 2, False), (128, 256, 3, 1, False), (256, 64, 5, 2, True),
    (160, 96, 4, 2, False), (192, 192, 5, 2, True), (224, 32, 3, 1, True)])
def test_wgrad_bf16_channel_pairs(ca, cb, k, s, transpose):
    """tfc_conv2d_wgrad in bfloat16 — the build with row-major stages and transposing LDS reads takes every pair of wide
    tensors — against the definition G[t][a][b] = sum_{n, q} A[n, q s + t - k/2, a] B[n, q, b] in float64 on the same
    (bfloat16-rounded) inputs; a pixel count that is not a multiple of the 64-pixel stage."""
    from compression_amd.layers.functional import conv2d_wgrad
    torch.manual_seed(12)
    n, hb, wb = 2, 7, 9
    ha, wa = hb * s, wb * s
    a = torch.randn(n, ha, wa, ca).bfloat16()
    b = torch.randn(n, hb, wb, cb).bfloat16()
    got = conv2d_wgrad(a.cuda(), b.cuda(), (k, k), s, transpose).cpu().double()
    ad, bd = a.double(), b.double()
    pad = k // 2
    ap = torch.zeros(n, ha + 2 * k, wa + 2 * k, ca, dtype=torch.float64)
    ap[:, k:k + ha, k:k + wa] = ad
    want = torch.zeros(k, k, ca, cb, dtype=torch.float64)
    for ty in range(k):
        for tx in range(k):
            win = ap[:, k + ty - pad:k + ty - pad + hb * s:s, k + tx - pad:k + tx - pad + wb * s:s]
            want[ty, tx] = torch.einsum("nyxa,nyxb->ab", win, bd)
    if transpose:
        want = want.transpose(2, 3)
    assert got.shape == want.shape
    assert (got - want).abs().max() <= 1e-3 * want.abs().max()      # float32 accumulation of exact bfloat16 products


def scipy_same_zeros(x, kernel, stride, up):
    """The reference test's oracle (signal_conv_test.py:171-218: zero-insertion upsampling, scipy.signal
    correlate / convolve in `valid` mode, strided read-out) behind the `same_zeros` padding of
    signal_conv.py:663-690 / :778-847: float64, channels-last.  x [N,H,W,Cin], kernel [kh,kw,Cin,Cout]."""
    import scipy.signal
    n, h, w, cin = x.shape
    kh, kw, _, cout = kernel.shape
    x = x.astype(np.float64)
    kernel = kernel.astype(np.float64)
    if up:
        xu = np.zeros((n, h * stride, w * stride, cin))            # extra_pad_end=True: length in * s
        xu[:, ::stride, ::stride] = x
        x, step = xu, 1
        kernel = kernel[::-1, ::-1]                                # transposed convolution = convolution
    else:
        step = stride
    pad = ((0, 0), (kh // 2, kh - 1 - kh // 2), (kw // 2, kw - 1 - kw // 2), (0, 0))
    if up:   # the convolution's window is anchored at the other end
        pad = ((0, 0), (kh - 1 - kh // 2, kh // 2), (kw - 1 - kw // 2, kw // 2), (0, 0))
    xp = np.pad(x, pad)
    out = np.empty((n, x.shape[1], x.shape[2], cout))
    for b in range(n):
        for co in range(cout):
            out[b, :, :, co] = scipy.signal.correlate(xp[b], kernel[:, :, :, co], mode="valid")[:, :, 0]
    return out[:, ::step, ::step]


@pytest.mark.parametrize("label,shape,kshape,stride,up", [
    ("analysis 5x5 192->192 /2", (2, 14, 18, 192), (5, 5, 192, 192), 2, False),
    ("synthesis 5x5 192->192 x2", (2, 9, 7, 192), (5, 5, 192, 192), 2, True),
    ("analysis 9x9 3->192 /4", (2, 29, 23, 3), (9, 9, 3, 192), 4, False),
    # whole 32-pixel tiles per output row: conv_image_direct_kernel (fragments straight from the image's rows)
    ("analysis 9x9 3->192 /4, rows of tiles", (2, 37, 128, 3), (9, 9, 3, 192), 4, False),
    ("analysis 9x9 3->192 /4, two tiles a row", (1, 16, 256, 3), (9, 9, 3, 192), 4, False),
    ("analysis 5x5 3->192 /2, rows of tiles", (2, 21, 64, 3), (5, 5, 3, 192), 2, False),
    ("analysis 5x5 3->192 /2, three tiles a row", (1, 6, 192, 3), (5, 5, 3, 192), 2, False),
    ("synthesis 9x9 192->3 x4", (1, 6, 7, 192), (9, 9, 192, 3), 4, True),
    ("synthesis 5x5 192->3 x2", (2, 8, 9, 192), (5, 5, 192, 3), 2, True),
    # several 8 x 32 input blocks with ragged edges (conv_up_phase_kernel)
    ("synthesis 9x9 192->3 x4, blocks", (2, 19, 70, 192), (9, 9, 192, 3), 4, True),
    ("synthesis 5x5 192->3 x2, blocks", (2, 17, 45, 192), (5, 5, 192, 3), 2, True),
    ("hyper 3x3 192->192 s1", (1, 8, 8, 192), (3, 3, 192, 192), 1, False),
])
def test_bf16_model_layer_shapes_against_scipy(label, shape, kshape, stride, up):
    """bf16 at the channel counts / kernels / strides of the two models' layers, against the scipy oracle of the
    reference's own test, in the reference's style: small integer inputs and kernels, so every product and
    partial sum is exact in fp32 and the only rounding is the output's (the few-channel transposed layer keeps
    its per-tap products in fp32 for that reason)."""
    from compression_amd.layers import conv2d_down, conv2d_up
    rng = np.random.default_rng(len(label))
    x = rng.integers(0, 8, shape).astype(np.float32)
    ker = rng.integers(-3, 4, kshape).astype(np.float32)
    want = scipy_same_zeros(x, ker, stride, up)
    fn = conv2d_up if up else conv2d_down
    y = fn(torch.from_numpy(x).bfloat16().cuda(), torch.from_numpy(ker), None, stride).float().cpu().numpy()
    assert y.shape == want.shape, label
    tol = 2.0 ** -8
    assert np.max(np.abs(y - want) - tol * np.abs(want)) <= 0.51 * tol, label   # exact small sums, 1/2 ulp elsewhere


@pytest.mark.parametrize("up,hw,cout", [(True, (24, 64), 192), (True, (32, 64), 128), (False, (64, 64), 192)])
@pytest.mark.parametrize("inverse", [False, True])
def test_gdn_as_the_activation_is_one_kernel(up, hw, cout, inverse):
    """SignalConv2D(activation=GDN) (signal_conv.py:948-950 applying gdn.py:371-421, the way bls2017.py:61-91 builds its
    transforms): where the third-generation kernel takes the layer the activation is its epilogue — same values as the
    convolution followed by the GDN kernel (both work on the bfloat16-rounded convolution output; one bfloat16 unit of
    the result at most), and no GDN kernel is launched."""
    import ctypes as C
    from compression_amd import _lib, layers
    torch.manual_seed(3)
    gdn = layers.GDN(inverse=inverse)
    kw = dict(corr=False, strides_up=2) if up else dict(corr=True, strides_down=2)
    conv = layers.SignalConv2D(cout, (5, 5), padding="same_zeros", in_channels=192, use_bias=True, activation=gdn, **kw).cuda()
    plain = layers.SignalConv2D(cout, (5, 5), padding="same_zeros", in_channels=192, use_bias=True, activation=None, **kw).cuda()
    x = (torch.randn(3, hw[0], hw[1], 192, device="cuda") * 2).to(torch.bfloat16)
    conv.fuse_gdn_activation = True
    with torch.no_grad():
        conv.build(192, x.device)
        conv.bias.normal_()
        gdn.build(cout, x.device)
        gdn.reparam_gamma.add_(torch.rand_like(gdn.reparam_gamma) * 0.05)       # off-diagonal weights too
        gdn.invalidate_kernel_cache()
        plain.load_state_dict({k: v for k, v in conv.state_dict().items() if not k.startswith("activation")}, strict=False)
        lib = _lib.lib()
        lib.tfc_profile_enable(1)
        y = conv(x)
        torch.cuda.synchronize()
        ms, n = C.c_double(), C.c_int64()
        lib.tfc_profile_query(b"gdn_forward", C.byref(ms), C.byref(n))
        lib.tfc_profile_enable(0)
        assert n.value == 0, "the GDN kernel ran behind a convolution that applies it itself"
        want = gdn(plain(x))
    assert y.shape == want.shape
    err = (y.float() - want.float()).abs()
    tol = want.float().abs() * 2.0 ** -7 + 1e-6
    assert bool((err <= tol).all()), float((err - tol).max())


@pytest.mark.parametrize("hw,batch", [((64, 64), 3), ((96, 192), 2), ((66, 128), 1), ((64, 72), 2)])
def test_image_side_layer_applies_its_gdn_itself(hw, batch):
    """bmshj2018's / bls2017's first analysis layer, SignalConv2D(192, (5, 5), corr=True, strides_down=2, activation=GDN)
    on a three-channel image (bls2017.py:66-69): conv_image_gdn_kernel writes the normalised activations — same values
    as conv_image_kernel followed by the GDN kernel (both normalise the bfloat16-rounded convolution output), all of
    the map including its zero-padded border, and no GDN launch.  A width the fused kernel does not take (output rows
    that are not a multiple of 32 pixels) goes through the two kernels."""
    import ctypes as C
    from compression_amd import _lib, layers
    torch.manual_seed(5)
    gdn = layers.GDN()
    kw = dict(corr=True, strides_down=2, padding="same_zeros", in_channels=3, use_bias=True)
    conv = layers.SignalConv2D(192, (5, 5), activation=gdn, **kw).cuda()
    plain = layers.SignalConv2D(192, (5, 5), activation=None, **kw).cuda()
    x = torch.rand(batch, hw[0], hw[1], 3, device="cuda").mul(255).to(torch.bfloat16)
    assert conv.fuse_gdn_image
    with torch.no_grad():
        conv.build(3, x.device)
        conv.bias.normal_()
        gdn.build(192, x.device)
        gdn.reparam_gamma.add_(torch.rand_like(gdn.reparam_gamma) * 0.05)
        gdn.invalidate_kernel_cache()
        plain.load_state_dict({k: v for k, v in conv.state_dict().items() if not k.startswith("activation")}, strict=False)
        lib = _lib.lib()
        lib.tfc_profile_enable(1)
        y = conv(x)
        torch.cuda.synchronize()
        ms, n = C.c_double(), C.c_int64()
        lib.tfc_profile_query(b"gdn_forward", C.byref(ms), C.byref(n))
        lib.tfc_profile_enable(0)
        fused = (hw[1] // 2) % 32 == 0
        assert n.value == (0 if fused else 1)
        want = gdn(plain(x))
    assert y.shape == want.shape == (batch, (hw[0] + 1) // 2, hw[1] // 2, 192)
    if fused:
        assert torch.equal(y, want)       # bias and beta are added to the finished sums, rounding as in the two kernels
    err = (y.float() - want.float()).abs()
    tol = want.float().abs() * 2.0 ** -7 + 1e-6
    assert bool((err <= tol).all()), float((err - tol).max())
    assert float(want.float().abs().max()) > 0.1


def test_keyed_weights_are_packed_once_per_value():
    """tfc_conv2d_weights_key (include/tfc_hip.h): the fragments of a keyed weight value are packed on the first call and
    reused — shown by overwriting the float32 kernel behind the library's back: the keyed call keeps the old value's
    output until the key is dropped; an unkeyed call always sees the tensor.  The layer keys its kernel by the
    parameters' version counters, so an in-place update under no_grad is a new value."""
    from compression_amd import _lib, layers
    from compression_amd.layers import conv2d_down, conv2d_up
    torch.manual_seed(11)
    lib = _lib.lib()
    key = (1 << 40) + 12345
    for fn, shape, kshape, stride in ((conv2d_down, (2, 32, 64, 192), (5, 5, 192, 192), 2),
                                      (conv2d_up, (2, 16, 32, 192), (5, 5, 192, 3), 2),
                                      (conv2d_down, (1, 64, 128, 3), (5, 5, 3, 192), 2)):
        x = torch.randn(shape, device="cuda").to(torch.bfloat16)
        w1 = torch.randn(kshape, device="cuda") / 30
        w2 = torch.randn(kshape, device="cuda") / 30
        w = w1.clone()
        y1 = fn(x, w, None, stride)
        y2 = fn(x, w2, None, stride)
        assert not torch.equal(y1, y2)
        assert torch.equal(fn(x, w, None, stride, weights_key=key), y1)
        w.copy_(w2)
        torch.cuda.synchronize()
        assert torch.equal(fn(x, w, None, stride, weights_key=key), y1)       # the kept fragments
        assert torch.equal(fn(x, w, None, stride), y2)                        # no key: packed from the tensor
        assert lib.tfc_conv2d_drop_weights(key) == 0
        assert torch.equal(fn(x, w, None, stride, weights_key=key), y2)
        assert lib.tfc_conv2d_drop_weights(key) == 0
    conv = layers.SignalConv2D(192, (5, 5), corr=True, strides_down=2, padding="same_zeros", in_channels=192,
                               use_bias=True).cuda()
    x = torch.randn(2, 32, 64, 192, device="cuda").to(torch.bfloat16)
    with torch.no_grad():
        conv.build(192, x.device)
        a, b = conv(x), conv(x)
        assert torch.equal(a, b) and conv._inference_weights_key() != 0
        first = conv._inference_weights_key()
        for p in conv.parameters():
            p.mul_(2.0)
        c = conv(x)
        assert conv._inference_weights_key() != first
        want = conv2d_down(x, conv.kernel, conv.bias, 2)
        assert torch.equal(c, want) and not torch.equal(c, a)
        # a write through `.data` advances neither address nor version counter: the layer is told (weights_changed), or
        # runs with keyed weights off
        for p in conv.parameters():
            p.data.mul_(0.5)
        torch.cuda.synchronize()
        assert torch.equal(conv(x), c)                                        # documented caveat: the kept fragments
        conv.weights_changed()
        d = conv(x)
        assert torch.equal(d, conv2d_down(x, conv.kernel, conv.bias, 2)) and not torch.equal(d, c)
        conv.keyed_weights = False
        assert conv._inference_weights_key() == 0
        for p in conv.parameters():
            p.data.mul_(2.0)
        conv.invalidate_kernel_cache()         # (the rdft kernel cache is keyed the same way)
        assert torch.equal(conv(x), conv2d_down(x, conv.kernel, conv.bias, 2))


@pytest.mark.parametrize("label,shape,kshape,stride,up", [
    ("analysis 9x9 3->192 /4", (2, 24, 128, 3), (9, 9, 3, 192), 4, False),
    ("analysis 5x5 3->192 /2", (2, 14, 64, 3), (5, 5, 3, 192), 2, False),
    ("synthesis 9x9 192->3 x4", (2, 11, 37, 192), (9, 9, 192, 3), 4, True),
    ("synthesis 5x5 192->3 x2", (2, 9, 40, 192), (5, 5, 192, 3), 2, True),
])
@pytest.mark.parametrize("activation", [None, "relu"])
def test_image_side_kernels_bias_and_relu_against_scipy(label, shape, kshape, stride, up, activation):
    """The image-side kernels' own epilogues (conv_image_direct_kernel, conv_up_phase_kernel): bias per output channel
    and the fused ReLU (signal_conv.py:940-950), against the scipy oracle with small integers — every sum exact in
    float32, the only rounding the output's."""
    from compression_amd.layers import conv2d_down, conv2d_up
    rng = np.random.default_rng(len(label) + (7 if activation else 0))
    x = rng.integers(0, 8, shape).astype(np.float32)
    ker = rng.integers(-3, 4, kshape).astype(np.float32)
    bias = rng.integers(-20, 21, kshape[-1]).astype(np.float32)
    want = scipy_same_zeros(x, ker, stride, up) + bias
    if activation == "relu":
        want = np.maximum(want, 0.0)
    fn = conv2d_up if up else conv2d_down
    y = fn(torch.from_numpy(x).bfloat16().cuda(), torch.from_numpy(ker), torch.from_numpy(bias), stride,
           activation).float().cpu().numpy()
    assert y.shape == want.shape, label
    tol = 2.0 ** -8
    assert np.max(np.abs(y - want) - tol * np.abs(want)) <= 0.51 * tol, label
    if activation == "relu":
        assert (y >= 0).all() and (want == 0).any()


# ---- every 2-D configuration of the reference's own test, on the kernels (the CPU tier checks the same cases' pad / crop
# arithmetic around emulations: tests/test_signal_conv_cpu.py) -----------------------------------------------------------
import signal_conv_cases as cases  # noqa: E402


def _run_layer_gpu(kernel, x_nchw, dtype=torch.float32, **kw):
    from compression_amd import layers
    layer = layers.SignalConv2D(kernel.shape[-1], kw.pop("kernel_support"), kernel_parameter=torch.from_numpy(kernel).cuda(), **kw)
    with torch.no_grad():
        y = layer(torch.from_numpy(np.moveaxis(x_nchw, 1, -1).copy()).to(dtype).cuda())
    return np.moveaxis(y.float().cpu().numpy(), -1, 1), layer


@pytest.mark.parametrize("case", [c for c in cases.valid_cases()
                                  if cases.is_implemented(c["kernel_support"], c["corr"], c["strides_up"], c["channel_separable"], c["filters"])],
                         ids=lambda c: "-".join(f"{k[:2]}{v}" for k, v in c.items()))
def test_reference_valid_cases_against_scipy(case):
    """signal_conv_test.py:224-260 `run_valid` (padding="valid", the reference's default): small integers, float32 —
    every product and sum exact, so the layer must equal SciPy exactly (the reference allows 1e-3)."""
    case = dict(case)
    rng = np.random.default_rng(1)
    support, channels, filters = case.pop("input_support"), case.pop("channels"), case.pop("filters")
    x = rng.integers(0, 32, (1, channels) + support).astype(np.float32)
    kernel = rng.integers(0, 16, case["kernel_support"] + (channels, filters)).astype(np.float32)
    want = cases.scipy_convolve_valid(case["corr"], x, kernel, case["strides_down"], case["strides_up"],
                                      case["extra_pad_end"], case["channel_separable"])
    got, layer = _run_layer_gpu(kernel, x, padding="valid", activation=(lambda t: t) if case["use_bias"] else None, **case)
    assert got.shape == want.shape
    assert np.array_equal(got, want)


@pytest.mark.parametrize("case", [c for c in cases.same_cases() if cases.is_implemented(c["kernel_support"], c["corr"], c["strides_up"], False, 1)],
                         ids=lambda c: "-".join(f"{k[:2]}{v}" for k, v in c.items()))
def test_reference_same_cases_identity_kernels(case):
    """signal_conv_test.py:262-315 `run_same`: `same_zeros` (explicit and pre-padded) and `same_reflect`."""
    case = dict(case)
    support = case.pop("input_support")
    x = np.arange(np.prod(support), dtype=np.float32).reshape((1, 1) + support)
    got, _ = _run_layer_gpu(cases.identity_kernel(case["kernel_support"], case["corr"]), x, **case)
    want = x
    if not all(s == 1 for s in case["strides_up"]):
        want = cases.numpy_upsample(want, case["strides_up"], case["extra_pad_end"])
    want = want[(slice(None), slice(None)) + tuple(slice(None, None, s) for s in case["strides_down"])]
    assert got.shape == want.shape
    assert np.array_equal(got, want)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("corr,ks,cin,cout", [(True, (3, 3), 3, 2), (False, (5, 3), 3, 2), (True, (5, 5), 192, 192),
                                              (False, (4, 3), 16, 8)])
def test_same_reflect_general_kernel_gpu(corr, ks, cin, cout, dtype):
    """`same_reflect` with a general kernel at small and model widths: the tfc_pad2d pre-pad + the `valid` path,
    against reflect-padded SciPy (small integers: exact in float32; bfloat16 rounds the output only)."""
    if not cases.is_implemented(ks, corr, (1, 1), False, cout):
        pytest.skip("not implemented by the reference either")
    rng = np.random.default_rng(2)
    x = rng.integers(0, 8, (2, cin, 9, 34)).astype(np.float32)
    kernel = rng.integers(-3, 4, ks + (cin, cout)).astype(np.float32)
    got, _ = _run_layer_gpu(kernel, x, dtype=dtype, kernel_support=ks, corr=corr, padding="same_reflect")
    want = cases.same_reflect_oracle(x, kernel, ks, corr)
    assert got.shape == want.shape
    if dtype == torch.float32:
        assert np.array_equal(got, want)
    else:
        tol = 2.0 ** -8
        assert np.max(np.abs(got - want) - tol * np.abs(want)) <= 0.51 * tol


def test_pad2d_kernel_against_numpy():
    from compression_amd.layers import functional
    rng = np.random.default_rng(3)
    for shape, ph, pw in (((2, 5, 7, 3), (2, 1), (0, 3)), ((1, 9, 6, 192), (3, 3), (2, 2)), ((3, 4, 4, 8), (0, 0), (1, 0))):
        for dtype in (torch.float32, torch.bfloat16):
            x = torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).to(dtype).cuda()
            for reflect in (False, True):
                with torch.no_grad():
                    y = functional.pad2d(x, ph, pw, reflect=reflect)
                want = np.pad(x.float().cpu().numpy(), ((0, 0), ph, pw, (0, 0)), mode="reflect" if reflect else "constant")
                assert np.array_equal(y.float().cpu().numpy(), want), (shape, ph, pw, dtype, reflect)
    # with a gradient wanted: the differentiable tensor ops, same values
    x = torch.randn(1, 5, 6, 4, device="cuda", requires_grad=True)
    y = functional.pad2d(x, (2, 1), (1, 2), reflect=True)
    assert np.array_equal(y.detach().cpu().numpy(), np.pad(x.detach().cpu().numpy(), ((0, 0), (2, 1), (1, 2), (0, 0)), mode="reflect"))
    y.sum().backward()
    assert x.grad is not None and float(x.grad.sum()) == y.numel()


def test_default_argument_layer_on_the_device():
    """`SignalConv2D(filters, k)` with default arguments (padding="valid", convolution, rdft kernel) constructs, runs, and
    matches SciPy; gradients reach its parameters through the pad / crop."""
    from compression_amd import layers
    torch.manual_seed(0)
    layer = layers.SignalConv2D(4, 3).cuda()
    x = torch.randn(2, 8, 9, 3, device="cuda")
    with torch.no_grad():
        y = layer(x)
    assert tuple(y.shape) == (2, 6, 7, 4)
    want = cases.scipy_convolve_valid(False, np.moveaxis(x.cpu().numpy(), -1, 1), layer.kernel.detach().cpu().numpy(),
                                      (1, 1), (1, 1), True, False)
    assert np.max(np.abs(np.moveaxis(y.cpu().numpy(), -1, 1) - want)) <= 1e-5
    layer(x).square().sum().backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in layer.parameters())


F32_SPLIT_CASES = [  # up, n, h, w, cin, cout, k, s — layer shapes of the models, float32
    (False, 2, 64, 96, 192, 192, 5, 2),      # third-generation kernel, float32 output
    (False, 1, 33, 47, 192, 192, 5, 2),      # second generation (ragged map)
    (False, 2, 32, 48, 192, 192, 3, 1),
    (True, 2, 32, 48, 192, 192, 5, 2),       # transposed: four phase groups
    (True, 1, 9, 11, 128, 128, 5, 2),
    (True, 1, 8, 7, 64, 4, 5, 2),            # few output channels: one partly filled column tile
    (False, 1, 20, 20, 16, 40, 4, 2),
]


@pytest.mark.parametrize("case", F32_SPLIT_CASES)
def test_float32_layers_on_the_bfloat16_matrix_cores(case, monkeypatch):
    """Round 6: a float32 SignalConv2D runs as ONE bfloat16 convolution over six times the input channels — three bfloat16
    planes per operand (a float32 is their exact sum), the six products above 2^-25 — with float32 accumulation and
    output (conv_split_x_kernel).  Both it and the float32-MFMA kernel it replaces (TFC_CONV_F32=native) are held to a
    float64 evaluation of the layer's definition (signal_conv.py:663-690, 778-847) at 4e-6 of the largest output — float32
    rounding noise over K = 25 x 192 products — and to each other at twice that."""
    from compression_amd.layers import conv2d_down, conv2d_up
    up, n, h, w, cin, cout, k, s = case
    torch.manual_seed(11)
    x = torch.randn(n, h, w, cin)
    ker = torch.randn(k, k, cin, cout) / np.sqrt(k * k * cin)
    bias = torch.randn(cout)
    fn = conv2d_up if up else conv2d_down
    want = (ref_up if up else ref_down)(x.double(), ker.double(), bias.double(), s, False)
    got = {}
    for mode in ("split", "native"):
        monkeypatch.setenv("TFC_CONV_F32", mode)
        got[mode] = fn(x.cuda(), ker, bias, s).cpu().double()
        assert got[mode].shape == want.shape
    scale = max(1.0, want.abs().max().item())
    for mode in got:
        assert (got[mode] - want).abs().max().item() <= 4e-6 * scale, mode
    assert (got["split"] - got["native"]).abs().max().item() <= 8e-6 * scale
