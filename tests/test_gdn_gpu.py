"""GPU tier: fused GDN/IGDN forward vs an fp64 numpy evaluation of
python/layers/gdn.py:371-421 (tolerance 1e-5 absolute on the float32 path — the
bar BASELINE.json states; bf16 path: bf16 rounding of inputs/gamma/outputs)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def ref_gdn(x, beta, gamma, inverse, rectify, alpha, eps):
    x = x.astype(np.float64)
    if rectify:
        x = np.maximum(x, 0)
    u = np.abs(x) if alpha == 1 else x * x
    n = u @ gamma.astype(np.float64) + beta.astype(np.float64)
    if eps == 0.5:
        n = np.sqrt(n)
    return x * n if inverse else x / n


def params(C, seed):
    g = torch.Generator().manual_seed(seed)
    beta = 1 + 0.1 * torch.rand(C, generator=g)
    gamma = 0.1 * torch.eye(C) + 0.01 * torch.rand(C, C, generator=g)
    return beta, gamma


@pytest.mark.parametrize("C", [32, 128, 192])
@pytest.mark.parametrize("inverse,rectify,alpha,eps", [
    (False, False, 1, 1), (True, False, 1, 1), (False, False, 2, 0.5), (True, True, 2, 0.5),
    (False, True, 1, 1)])
def test_gdn_f32(C, inverse, rectify, alpha, eps):
    from compression_amd.layers import gdn_forward
    torch.manual_seed(3)
    x = torch.randn(5, 7, 9, C)          # 315 pixels: exercises the ragged last tile
    beta, gamma = params(C, 1)
    y = gdn_forward(x.cuda(), beta, gamma, inverse, rectify, alpha, eps).cpu().numpy()
    want = ref_gdn(x.numpy(), beta.numpy(), gamma.numpy(), inverse, rectify, alpha, eps)
    err = np.abs(y - want)
    print(f"gdn_f32 C={C} inverse={inverse} alpha={alpha} eps={eps}: max |y - want| = {err.max():.3e} "
          f"(max |want| = {np.abs(want).max():.3f})")
    if not inverse:
        # GDN proper (BASELINE.json: "within 1e-5 on GDN float outputs"): ABSOLUTE, every element
        assert err.max() <= 1e-5
    else:
        # IGDN multiplies by the norm (outputs up to ~30 on these inputs; one float32 ulp there is 2e-6): 1e-5 per
        # element, relative to that element where it exceeds 1 — never to the tensor's maximum
        assert np.max(err / np.maximum(1.0, np.abs(want))) <= 1e-5


@pytest.mark.parametrize("C", [32, 64, 96, 160, 192, 224, 256])     # (an odd number of 32-channel tiles: a last output line of 64 bytes)
@pytest.mark.parametrize("inverse", [False, True])
def test_gdn_bf16(C, inverse):
    from compression_amd.layers import gdn_forward
    torch.manual_seed(4)
    x = torch.randn(3, 33, C).bfloat16()
    beta, gamma = params(C, 2)
    y = gdn_forward(x.cuda(), beta, gamma, inverse).float().cpu().numpy()
    # reference on the bf16-rounded inputs the kernel sees
    want = ref_gdn(x.float().numpy(), beta.numpy(), gamma.bfloat16().float().numpy(), inverse, False, 1, 1)
    assert np.max(np.abs(y - want) / (np.abs(want) + 1e-3)) <= 2 ** -7   # one bf16 ulp of slack on the output


@pytest.mark.parametrize("C", [96, 192])
@pytest.mark.parametrize("inverse,rectify,alpha,eps", [
    (False, True, 1, 1), (True, False, 2, 0.5), (False, False, 2, 1), (True, True, 1, 0.5), (False, True, 1.5, 0.7)])
def test_gdn_bf16_every_forward_variant(C, inverse, rectify, alpha, eps):
    """The bfloat16 forward kernel's other builds (rectify / alpha = 2 / epsilon = 1/2: the general epilogue without the
    next-tile prefetch at 192 channels; learned exponents: the GEN build) write through the same whole-line store path."""
    from compression_amd.layers import gdn_forward
    torch.manual_seed(9)
    x = torch.randn(5, 41, C).bfloat16()
    beta, gamma = params(C, 3)
    y = gdn_forward(x.cuda(), beta, gamma, inverse=inverse, rectify=rectify, alpha=alpha, epsilon=eps).float().cpu().numpy()
    xf = x.float().numpy().astype(np.float64)
    if rectify:
        xf = np.maximum(xf, 0)
    u = np.abs(xf) ** alpha
    # (the kernel feeds |x|^alpha to the matrix cores as bfloat16, gamma likewise)
    n = u @ gamma.bfloat16().float().numpy().astype(np.float64) + beta.numpy().astype(np.float64)
    want = xf * n ** eps if inverse else xf / n ** eps
    assert np.max(np.abs(y - want) / (np.abs(want) + 1e-2)) <= 2 ** -6


def test_gdn_bf16_whole_line_stores_both_ways():
    """The forward kernel stores y as whole 128-byte lines, non-temporal where x + y exceed 128 MB (gdn_common.h): a tensor
    on each side of that line, ragged in pixels (not a multiple of the 32-pixel tile), gives the values of the
    pixel-by-pixel float64 evaluation on a sample of its rows — first, last and the rows around a tile boundary."""
    from compression_amd.layers import gdn_forward
    C = 192
    beta, gamma = params(C, 2)
    gen = torch.Generator().manual_seed(11)
    for pixels in (1000 + 7, 200_000 + 13):          # 0.8 MB and 154 MB of x + y
        x = torch.randn(pixels, C, generator=gen).bfloat16()
        y = gdn_forward(x.cuda(), beta, gamma, False).float().cpu()
        rows = torch.tensor([0, 1, 31, 32, 33, pixels // 2, pixels - 34, pixels - 33, pixels - 2, pixels - 1])
        want = ref_gdn(x[rows].float().numpy(), beta.numpy(), gamma.bfloat16().float().numpy(), False, False, 1, 1)
        got = y[rows].numpy()
        assert np.max(np.abs(got - want) / (np.abs(want) + 1e-3)) <= 2 ** -7, pixels
        # and every row was written: nothing of the output is left at the allocator's fill
        assert torch.isfinite(y).all() and (y.abs().sum(dim=1) > 0).all()


def test_gdn_closed_form():
    # gdn_test.py:42-88: with beta = 1, gamma = .1 I the layer is x / (1 + .1 |x|).
    from compression_amd.layers import gdn_forward
    C = 32
    x = torch.rand(2, 4, 4, C) * 2 - 1
    y = gdn_forward(x.cuda(), torch.ones(C), 0.1 * torch.eye(C)).cpu()
    assert torch.allclose(y, x / (1 + 0.1 * x.abs()), atol=1e-6)
    yi = gdn_forward(x.cuda(), torch.ones(C), 0.1 * torch.eye(C), inverse=True).cpu()
    assert torch.allclose(yi, x * (1 + 0.1 * x.abs()), atol=1e-6)


def test_gdn_config3_full_size_property():
    """BASELINE config 3: 256x192x32x32 bf16.  GDN followed by IGDN with the norm
    recomputed is not an identity, so check linearity in the scale instead:
    IGDN(GDN(x)) uses different norms; the size-independent property used here is
    permutation equivariance over pixels and agreement with a sampled fp64 oracle."""
    from compression_amd.layers import gdn_forward
    torch.manual_seed(3)
    x = torch.randn(256 * 32 * 32, 192, dtype=torch.bfloat16, device="cuda")
    beta, gamma = params(192, 3)
    y = gdn_forward(x, beta, gamma)
    perm = torch.randperm(x.shape[0], device="cuda")
    y2 = gdn_forward(x[perm].contiguous(), beta, gamma)
    assert torch.equal(y[perm], y2)
    idx = torch.randint(0, x.shape[0], (512,), device="cuda")
    want = ref_gdn(x[idx].float().cpu().numpy(), beta.numpy(), gamma.bfloat16().float().numpy(),
                   False, False, 1, 1)
    got = y[idx].float().cpu().numpy()
    assert np.max(np.abs(got - want) / (np.abs(want) + 1e-3)) <= 2 ** -7


def ref_gdn_grads(x, g, beta, gamma, inverse, rectify, alpha, eps):
    """fp64 autograd through the formula of python/layers/gdn.py:371-421 (the reference has
    no hand-written gradient: TF autodiff differentiates exactly these ops)."""
    x = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    beta = torch.tensor(beta, dtype=torch.float64, requires_grad=True)
    gamma = torch.tensor(gamma, dtype=torch.float64, requires_grad=True)
    xx = torch.relu(x) if rectify else x
    u = xx.abs() if alpha == 1 else xx * xx
    n = u @ gamma + beta
    if eps == 0.5:
        n = n.sqrt()
    y = xx * n if inverse else xx / n
    y.backward(torch.tensor(g, dtype=torch.float64))
    return x.grad.numpy(), beta.grad.numpy(), gamma.grad.numpy()


@pytest.mark.parametrize("C", [32, 96, 192])
@pytest.mark.parametrize("inverse,rectify,alpha,eps", [
    (False, False, 1, 1), (True, False, 1, 1), (False, False, 2, 0.5), (True, True, 2, 0.5),
    (False, True, 1, 1), (True, False, 1, 0.5)])
def test_gdn_backward_f32(C, inverse, rectify, alpha, eps):
    from compression_amd.layers import gdn_backward
    torch.manual_seed(5)
    x = torch.randn(3, 7, 11, C)          # 231 pixels: ragged tile and ragged LDS stage
    g = torch.randn(3, 7, 11, C)
    beta, gamma = params(C, 1)
    dx, dbeta, dgamma = gdn_backward(x.cuda(), g.cuda(), beta, gamma, inverse, rectify, alpha, eps)
    wx, wb, wg = ref_gdn_grads(x.reshape(-1, C).numpy(), g.reshape(-1, C).numpy(), beta.numpy(),
                               gamma.numpy(), inverse, rectify, alpha, eps)
    assert np.max(np.abs(dx.cpu().numpy().reshape(-1, C) - wx)) <= 1e-5 * max(1.0, np.max(np.abs(wx)))
    assert np.max(np.abs(dbeta.cpu().numpy() - wb)) <= 1e-4 * max(1.0, np.max(np.abs(wb)))
    assert np.max(np.abs(dgamma.cpu().numpy() - wg)) <= 1e-4 * max(1.0, np.max(np.abs(wg)))


@pytest.mark.parametrize("C", [64, 160, 192, 256])     # <= 192: fused kernel; 256: three passes
@pytest.mark.parametrize("inverse,rectify,alpha,eps", [
    (False, False, 1, 1), (True, False, 1, 1), (False, True, 2, 0.5), (True, True, 1, 0.5),
    (False, False, 2, 1)])
def test_gdn_backward_bf16(C, inverse, rectify, alpha, eps):
    from compression_amd.layers import gdn_backward
    torch.manual_seed(6)
    x = torch.randn(5, 41, C).bfloat16()
    g = torch.randn(5, 41, C).bfloat16()
    beta, gamma = params(C, 2)
    dx, dbeta, dgamma = gdn_backward(x.cuda(), g.cuda(), beta, gamma, inverse, rectify, alpha, eps)
    wx, wb, wg = ref_gdn_grads(x.float().reshape(-1, C).numpy(), g.float().reshape(-1, C).numpy(),
                               beta.numpy(), gamma.bfloat16().float().numpy(), inverse, rectify, alpha, eps)
    # bf16 storage of T, R and dx: a few bf16 ulps relative to the tensor scale
    assert np.max(np.abs(dx.float().cpu().numpy().reshape(-1, C) - wx)) <= 2 ** -6 * np.max(np.abs(wx))
    assert np.max(np.abs(dbeta.cpu().numpy() - wb)) <= 2 ** -6 * np.max(np.abs(wb)) + 0.05
    assert np.max(np.abs(dgamma.cpu().numpy() - wg)) <= 2 ** -6 * np.max(np.abs(wg)) + 0.05


def test_gdn_module_autograd():
    """GDN module end to end: loss.backward() reaches x and the reparameterised beta / gamma."""
    from compression_amd.layers import GDN
    torch.manual_seed(7)
    layer = GDN(64).cuda()
    x = torch.randn(2, 6, 6, 64, device="cuda", requires_grad=True)
    layer(x).square().sum().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all() and x.grad.abs().sum() > 0
    grads = [p.grad for p in layer.parameters()]
    assert grads and all(g is not None and torch.isfinite(g).all() for g in grads)
    assert any(g.abs().sum() > 0 for g in grads)


def ref_gdn_general(x, beta, gamma, inverse, rectify, alpha, eps):
    """gdn.py:377-416 with the tf.pow paths (`inputs ** alpha`, `norm_pool ** epsilon`), in float64."""
    x = x.astype(np.float64)
    if rectify:
        x = np.maximum(x, 0)
    with np.errstate(invalid="ignore"):
        u = np.power(x, alpha)
    n = u @ gamma.astype(np.float64) + beta.astype(np.float64)
    n = np.power(n, eps)
    return x * n if inverse else x / n


@pytest.mark.parametrize("C", [32, 96, 192])
@pytest.mark.parametrize("inverse,rectify,alpha,eps", [
    (False, True, 1.5, 0.7), (True, True, 1.25, 0.3), (True, False, 3.0, 1.0), (True, False, 2.0, 0.8),
    (False, True, 1.0, 0.25)])
def test_gdn_general_exponents_f32(C, inverse, rectify, alpha, eps):
    """General (learned-style) alpha / epsilon: the forward kernel's tf.pow variant against float64.
    Without rectify only integer alphas are real for negative inputs (tf.pow, numpy.power alike); with an odd
    one the norm pool can be negative or near zero, so that case multiplies (IGDN) instead of dividing."""
    from compression_amd.layers import gdn_forward
    torch.manual_seed(5)
    x = torch.randn(4, 9, 7, C)
    beta, gamma = params(C, 6)
    y = gdn_forward(x.cuda(), beta, gamma, inverse, rectify, alpha, eps).cpu().numpy()
    want = ref_gdn_general(x.numpy(), beta.numpy(), gamma.numpy(), inverse, rectify, alpha, eps)
    assert np.isfinite(want).all()
    assert np.max(np.abs(y - want)) <= 1e-5 * max(1.0, np.max(np.abs(want)))


def test_gdn_general_negative_base_is_nan():
    """x ** 1.5 of a negative x is NaN (tf.pow): every output channel whose norm pool sees it is NaN."""
    from compression_amd.layers import gdn_forward
    C = 32
    x = torch.rand(2, 3, C) + 0.5
    x[0, 1, 4] = -0.75
    beta, gamma = params(C, 7)
    y = gdn_forward(x.cuda(), beta, gamma, False, False, 1.5, 1.0).cpu()
    assert torch.isnan(y[0, 1]).all() and not torch.isnan(y[0, 0]).any() and not torch.isnan(y[1]).any()


def test_gdn_general_bf16():
    from compression_amd.layers import gdn_forward
    torch.manual_seed(8)
    C = 192
    x = torch.randn(3, 40, C).bfloat16()
    beta, gamma = params(C, 9)
    y = gdn_forward(x.cuda(), beta, gamma, False, True, 1.3, 0.6).float().cpu().numpy()
    xf = np.maximum(x.float().numpy().astype(np.float64), 0)
    u = torch.from_numpy(np.power(xf, 1.3)).bfloat16().double().numpy()      # the kernel rounds u to bf16 for the MFMA
    n = u @ gamma.bfloat16().double().numpy() + beta.double().numpy()
    want = xf / np.power(n, 0.6)
    assert np.max(np.abs(y - want) / (np.abs(want) + 1e-3)) <= 2 ** -6


def test_gdn_layer_learned_alpha_epsilon():
    """alpha_parameter=None / epsilon_parameter=None: learned scalars (gdn.py:345-369).  Gradients of
    x, beta, gamma, alpha, epsilon against float64 autograd of the same formula; the no-grad call of the
    same layer runs the forward kernel and agrees with the composite."""
    import compression_amd as tfc
    torch.manual_seed(10)
    C = 64
    layer = tfc.layers.GDN(rectify=True, alpha_parameter=None, epsilon_parameter=None,
                           alpha_initializer=lambda: torch.tensor(1.4), epsilon_initializer=lambda: torch.tensor(0.6),
                           num_channels=C).cuda()
    assert abs(float(layer.alpha.detach()) - 1.4) < 1e-5 and abs(float(layer.epsilon.detach()) - 0.6) < 1e-5
    names = {n for n, _ in layer.named_parameters()}
    assert {"reparam_alpha", "reparam_epsilon", "reparam_beta", "reparam_gamma"} <= names
    x = (torch.rand(2, 5, 6, C, device="cuda") * 2 - 0.5).requires_grad_(True)
    w = torch.randn(2, 5, 6, C, device="cuda")
    y = layer(x)
    (y * w).sum().backward()
    # float64 reference
    xd = x.detach().double().requires_grad_(True)
    a = layer.alpha.detach().double().requires_grad_(True)
    e = layer.epsilon.detach().double().requires_grad_(True)
    b = layer.beta.detach().double().requires_grad_(True)
    g = layer.gamma.detach().double().requires_grad_(True)
    xr = torch.relu(xd)
    yd = xr / torch.pow(torch.pow(xr, a) @ g + b, e)
    (yd * w.double()).sum().backward()
    assert torch.allclose(y.double(), yd, atol=1e-5)
    assert torch.allclose(x.grad.double(), xd.grad, atol=1e-4, rtol=1e-4)
    # parameters: chain through the reparameterisation  value = variable^2 - pedestal
    for rep, grad in ((layer.reparam_alpha, a.grad), (layer.reparam_epsilon, e.grad)):
        want = grad * 2 * rep.detach().double()
        assert torch.allclose(rep.grad.double(), want, rtol=1e-3, atol=1e-4), (rep.grad, want)
    with torch.no_grad():
        y_kernel = layer(x.detach())
    assert torch.allclose(y_kernel, y.detach(), atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("inverse", [False, True])
def test_prepared_parameters_give_the_same_outputs(dtype, inverse):
    """tfc_gdn_params_create + tfc_gdn_forward_prepared (the parameter image built once) == tfc_gdn_forward, bit for
    bit; the layer takes that path under no_grad and rebuilds the image when its variables change."""
    from compression_amd.layers import functional
    torch.manual_seed(7)
    C = 96
    x = torch.randn(3, 17, 19, C, device="cuda").to(dtype)
    beta = 1 + 0.1 * torch.rand(C, device="cuda")
    gamma = 0.1 * torch.eye(C, device="cuda") + 0.01 * torch.rand(C, C, device="cuda")
    want = functional.gdn_forward(x, beta, gamma, inverse=inverse)
    prep = functional.GDNPrepared(beta, gamma, dtype)
    assert torch.equal(functional.gdn_forward(x, beta, gamma, inverse=inverse, prepared=prep), want)
    import compression_amd as tfc
    layer = tfc.GDN(inverse=inverse).cuda()
    y_grad = layer(x)                                   # builds the variables; autograd path
    with torch.no_grad():
        y0 = layer(x)
        assert torch.equal(y0, y_grad.detach())
        assert "prepared" in layer.__dict__["_value_cache"]
        layer.reparam_beta.mul_(1.5)                    # version bump: values and image are rebuilt
        y1 = layer(x)
    assert not torch.equal(y1, y0)
    assert torch.equal(y1, functional.gdn_forward(x, layer.beta, layer.gamma, inverse=inverse))
