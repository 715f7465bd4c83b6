"""GPU tier: compress/decompress of the entropy models (ports of
continuous_batched_test.py:103-242 and continuous_indexed_test.py) plus pipeline
parity: the byte strings equal the CPU oracle's for the same tables and symbols."""
import math

import numpy as np
import pytest
import torch

import compression_amd as tfc

pytestmark = pytest.mark.gpu


def test_instantiate_statelessly_and_tables():
    noisy = tfc.NoisyNormal(loc=0.25, scale=1.0)
    em = tfc.ContinuousBatchedEntropyModel(noisy, coding_rank=1, compression=True)
    assert em.compression and not em.stateless
    assert float(em.quantization_offset) == 0.25
    em2 = tfc.ContinuousBatchedEntropyModel(
        compression=True, stateless=True, coding_rank=1, prior_shape=noisy.batch_shape,
        cdf=em.cdf, cdf_offset=em.cdf_offset, quantization_offset=em.quantization_offset)
    assert em2.stateless and float(em2.quantization_offset) == 0.25
    with pytest.raises(RuntimeError):
        em2.prior
    assert em2.range_coder_precision == 12


def test_compression_consistent_with_quantization():
    torch.manual_seed(0)
    noisy = tfc.NoisyNormal(loc=0.25, scale=10.0)
    em = tfc.ContinuousBatchedEntropyModel(noisy, 1, compression=True)
    x = 0.25 + 10 * torch.randn(100)
    xq = em.quantize(x)
    for fused in (True, False):
        em.fused = fused
        xd = em.decompress(em.compress(x), [100])
        assert torch.equal(xd.cpu(), xq), fused


def test_pipeline_parity_with_oracle(port):
    """compress() == oracle(range coder) on the model's own tables and symbols."""
    torch.manual_seed(1)
    prior = tfc.NoisyNormal(loc=torch.linspace(-0.4, 0.4, 24), scale=torch.logspace(-1, 1.3, 24))
    em = tfc.ContinuousBatchedEntropyModel(prior, coding_rank=3, compression=True)
    y = torch.randn(5, 6, 7, 24) * torch.logspace(-1, 1.3, 24) * 1.5   # wider than the prior: escapes
    strings = em.compress(y)
    assert strings.shape == (5,)
    off = em.quantization_offset
    sym = torch.round(y - off).to(torch.int32).reshape(5, -1) - em.cdf_offset.repeat(42)
    want, _, _ = port.encode(em.cdf.numpy(), sym.numpy())
    assert [bytes(s) for s in strings] == want
    back = em.decompress(strings, [6, 7])
    assert torch.equal(back.cpu(), em.quantize(y))


@pytest.mark.parametrize("scale", [2.0 ** i for i in (-2, 0, 3, 7)])
def test_information_bounds(scale):
    torch.manual_seed(2)
    prior = tfc.NoisyNormal(loc=0.5, scale=scale)
    em = tfc.ContinuousBatchedEntropyModel(prior, coding_rank=1, compression=True)
    x = 0.5 + scale * torch.randn(1000000)
    _, bits_eval = em(x, training=False)
    _, bits_training = em(x, training=True)
    s = em.compress(x)
    bits_compressed = 8 * len(bytes(s[()]))
    # asymptotic bound; at 1e6 samples the two estimates fluctuate by ~1e-3 relative
    assert float(bits_training) > 0.998 * float(bits_eval)
    assert bits_compressed > float(bits_eval)
    if scale >= 64:
        assert abs(float(bits_training) - float(bits_eval)) <= 1e-4 * float(bits_eval)
    assert abs(bits_compressed - float(bits_eval)) <= 5e-3 * float(bits_eval)


def test_compression_works_after_serialization():
    torch.manual_seed(3)
    for loc, scale in ((0.5, 8.0), (0.0, 5.0)):
        noisy = tfc.NoisyNormal(loc=loc, scale=scale)
        em = tfc.ContinuousBatchedEntropyModel(noisy, 1, compression=True)
        assert (em._quantization_offset is None) == (loc == 0.0)
        config, weights, state = em.get_config(), em.get_weights(), em.state_dict()
        x = loc + scale * torch.randn(100)
        xq, xc = em.quantize(x), em.compress(x)
        em2 = tfc.ContinuousBatchedEntropyModel.from_config(config)
        em2.set_weights(weights)
        assert bytes(em2.compress(x)[()]) == bytes(xc[()])
        assert torch.equal(em2.decompress(xc, [100]).cpu(), xq)
        em3 = tfc.ContinuousBatchedEntropyModel.from_config(config)
        em3.load_state_dict(state)
        assert bytes(em3.compress(x)[()]) == bytes(xc[()])


def test_mixed_precision_bottleneck():
    noisy = tfc.NoisyNormal(loc=torch.tensor(0.5, dtype=torch.float64),
                            scale=torch.tensor(1.0, dtype=torch.float64), dtype=torch.float64)
    for dt in (torch.float16, torch.bfloat16):
        em = tfc.ContinuousBatchedEntropyModel(noisy, 1, compression=True, bottleneck_dtype=dt)
        assert em.prior.dtype == torch.float64
        x = torch.randn(2, 5).to(dt)
        xt, bits = em(x)
        s = em.compress(x)
        xh = em.decompress(s, (5,))
        assert xh.dtype == dt and xt.dtype == dt and bits.dtype == torch.float64 and bits.shape == (2,)
        assert (xh.cpu().float() - x.float()).abs().max() <= 0.5 + 2 ** -6
        assert torch.equal(xh.cpu(), em.quantize(x))
        assert (bits >= 0).all()


def test_dirac_priors():
    prior = tfc.NoisyNormal(loc=100.0 * torch.arange(16.0), scale=1e-10)
    em = tfc.ContinuousBatchedEntropyModel(prior, coding_rank=2, offset_heuristic=False, compression=True)
    assert em.cdf_offset.shape[0] == 16 and em.cdf.shape[0] <= 16 * 6
    em = tfc.ContinuousBatchedEntropyModel(prior, coding_rank=2, compression=True)
    x = (100.0 * torch.arange(16.0)).expand(3, 1000, 16).contiguous()
    _, bits_estimate = em(x, training=True)
    strings = em.compress(x)
    xd = em.decompress(strings, (1000,))
    assert (bits_estimate <= 16).all()
    assert all(8 * len(bytes(b)) <= 16 for b in strings.reshape(-1))
    assert (xd.cpu() - x).abs().max() <= 0.5


def test_noisy_deep_factorized_bottleneck(port):
    torch.manual_seed(5)
    em = tfc.EntropyBottleneck(8, compression=True)          # bls2017's entropy model, 8 filters
    y = torch.randn(2, 4, 4, 8) * 6
    s = em.compress(y)
    assert s.shape == (2,)
    yh = em.decompress(s, (4, 4))
    assert torch.equal(yh.cpu(), em.quantize(y))
    # byte parity with the oracle on the same tables
    off = em.quantization_offset
    sym = (torch.round(y - off) if off is not None else torch.round(y)).to(torch.int32)
    sym = sym.reshape(2, -1) - em.cdf_offset.repeat(16)
    want, _, _ = port.encode(em.cdf.numpy(), sym.numpy())
    assert [bytes(b) for b in s] == want


def _scale_fn(i):
    return torch.exp(math.log(0.11) + (math.log(256.0) - math.log(0.11)) / 63 * i)


def test_location_scale_indexed_roundtrip_and_parity(port):
    torch.manual_seed(6)
    em = tfc.LocationScaleIndexedEntropyModel(tfc.NoisyNormal, num_scales=64, scale_fn=_scale_fn,
                                              coding_rank=3, compression=True)
    idx = torch.randint(0, 64, (3, 8, 12, 16)).float()
    y = torch.randn(3, 8, 12, 16) * _scale_fn(idx) * 1.2
    for fused in (True, False):
        em.fused = fused
        s = em.compress(y, idx)
        assert s.shape == (3,)
        yh = em.decompress(s, idx)
        assert torch.equal(yh.cpu(), torch.round(y)), fused
    flat = idx.to(torch.int32)
    sym = torch.round(y).to(torch.int32) - em.cdf_offset[flat.long()]
    want, _, _ = port.encode(em.cdf.numpy(), sym.reshape(3, -1).numpy(), index=flat.reshape(3, -1).numpy())
    assert [bytes(b) for b in s] == want
    loc = torch.randn_like(y)
    s2 = em.compress(y, idx, loc=loc)
    assert torch.allclose(em.decompress(s2, idx, loc=loc.cuda()).cpu(), torch.round(y - loc) + loc)


def test_indexed_model_with_two_index_dimensions():
    """index_ranges=(a, b), channel_axis=-1 (continuous_indexed.py:272-289): the flattened index is
    computed on the device elementwise (integer tensordot has no HIP kernel), and coding_rank=0 keeps
    per-element bits."""
    torch.manual_seed(8)
    em = tfc.ContinuousIndexedEntropyModel(
        tfc.NoisyNormal, index_ranges=(4, 6), channel_axis=-1, coding_rank=1, compression=True,
        parameter_fns=dict(loc=lambda i: i[..., 0].float() - 1.5, scale=lambda i: torch.exp(0.4 * i[..., 1].float())))
    idx = torch.stack([torch.randint(0, 4, (5, 300)), torch.randint(0, 6, (5, 300))], dim=-1).float()
    loc = idx[..., 0] - 1.5
    y = loc + torch.randn(5, 300) * torch.exp(0.4 * idx[..., 1])
    s = em.compress(y, idx)
    assert s.shape == (5,)
    assert torch.equal(em.decompress(s, idx).cpu(), torch.round(y))
    flat = em._flatten_indexes(em._normalize_indexes(idx).cuda()).cpu()
    assert torch.equal(flat, (idx[..., 0] * 6 + idx[..., 1]).to(torch.int32))
    em0 = tfc.ContinuousIndexedEntropyModel(
        tfc.NoisyNormal, index_ranges=(4, 6), channel_axis=-1, coding_rank=0,
        parameter_fns=dict(loc=lambda i: i[..., 0].float() - 1.5, scale=lambda i: torch.exp(0.4 * i[..., 1].float())))
    _, bits = em0(y, idx, training=False)
    assert bits.shape == y.shape


def test_indexed_bits_close_to_rate():
    torch.manual_seed(7)
    em = tfc.LocationScaleIndexedEntropyModel(tfc.NoisyNormal, num_scales=64, scale_fn=_scale_fn,
                                              coding_rank=1, compression=True)
    idx = torch.randint(8, 56, (200000,)).float()
    y = torch.randn(200000) * _scale_fn(idx)
    _, bits = em(y, idx, training=False)
    n = 8 * len(bytes(em.compress(y, idx)[()]))
    assert float(bits) < n <= 1.01 * float(bits)


@pytest.mark.parametrize("num_filters,dtype", [((3, 3), torch.float32), ((3, 3, 3), torch.float32),
                                                ((5, 5), torch.float32), ((3, 3), torch.bfloat16)])
def test_fused_training_bottleneck_matches_torch_path(num_filters, dtype):
    """csrc/factorized_bits.hip against the op-by-op evaluation of continuous_batched.py:291-322
    (torch autograd through uniform_noise.py:117-156 / deep_factorized.py:166-194): same perturbed
    tensor, bits within 1e-5 relative (f32), gradients w.r.t. the input and every prior parameter."""
    from compression_amd.ops import bottleneck_ops
    torch.manual_seed(11)
    C = 48
    prior = tfc.NoisyDeepFactorized(batch_shape=(C,), num_filters=num_filters).cuda()
    with torch.no_grad():
        for prm in prior.parameters():
            prm.add_(0.3 * torch.randn_like(prm))
    y = (3.0 * torch.randn(3, 5, 7, C, device="cuda")).to(dtype).requires_grad_(True)
    noise = (torch.rand(3, 5, 7, C, device="cuda") - 0.5).to(dtype)
    w = torch.tensor([1.0, -2.0, 0.5], device="cuda")

    def reference():
        y_hat = y + noise
        lp = prior.log_prob(y_hat.to(torch.float32))
        return y_hat, lp.sum(dim=(1, 2, 3)) / -float(np.log(2.0))

    def fused():
        return bottleneck_ops.factorized_bits(y, prior.base, 3, noise)

    outs = []
    for fn in (reference, fused):
        y.grad = None
        prior.zero_grad()
        y_hat, bits = fn()
        ((bits * w).sum() + (y_hat.float() ** 2).sum() * 1e-3).backward()
        outs.append((y_hat.detach(), bits.detach(), y.grad.detach().float().clone(),
                     [p.grad.detach().clone() for p in prior.parameters()]))
    (yr, br, gr, pr), (yf, bf, gf, pf) = outs
    assert torch.equal(yr, yf)
    tol = 1e-5 if dtype == torch.float32 else 1e-5
    assert torch.allclose(br, bf, rtol=tol, atol=1e-3)
    gtol = 2e-4 if dtype == torch.float32 else 2e-2       # bf16: dy is stored in bf16
    assert torch.allclose(gr, gf, rtol=gtol, atol=gtol * gr.abs().max().item())
    for a, b in zip(pr, pf):
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-3 * max(a.abs().max().item(), 1e-3))


def test_entropy_model_training_call_uses_fused_path():
    """ContinuousBatchedEntropyModel(training=True): bits agree with the eval-free torch evaluation of
    the same perturbed tensor and gradients reach the prior."""
    torch.manual_seed(12)
    C = 32
    prior = tfc.NoisyDeepFactorized(batch_shape=(C,)).cuda()
    em = tfc.ContinuousBatchedEntropyModel(prior, coding_rank=3, compression=False)
    y = torch.randn(2, 6, 6, C, device="cuda", requires_grad=True)
    y_hat, bits = em(y, training=True)
    assert y_hat.shape == y.shape and bits.shape == (2,)
    assert (y_hat - y).abs().max() <= 0.5
    want = prior.log_prob(y_hat.detach()).sum(dim=(1, 2, 3)) / -float(np.log(2.0))
    assert torch.allclose(bits.detach(), want, rtol=1e-5, atol=1e-3)
    bits.sum().backward()
    assert y.grad is not None and all(p.grad is not None and torch.isfinite(p.grad).all() for p in prior.parameters())


@pytest.mark.parametrize("num_filters,dtype", [((3, 3), torch.float32), ((3, 3, 3), torch.float32),
                                                ((3, 3), torch.bfloat16)])
def test_fused_training_bottleneck_expected_grads(num_filters, dtype):
    """expected_grads=True (math_ops.py:157-216): the fused backward's finite-difference input gradient
    and its parameter gradients against math_ops.perturb_and_apply on the torch evaluation of log_prob."""
    from compression_amd.ops import bottleneck_ops, math_ops
    torch.manual_seed(13)
    C = 48
    prior = tfc.NoisyDeepFactorized(batch_shape=(C,), num_filters=num_filters).cuda()
    with torch.no_grad():
        for prm in prior.parameters():
            prm.add_(0.3 * torch.randn_like(prm))
    y = (3.0 * torch.randn(3, 5, 7, C, device="cuda")).to(dtype).requires_grad_(True)
    noise = (torch.rand(3, 5, 7, C, device="cuda") - 0.5).to(dtype)
    w = torch.tensor([1.0, -2.0, 0.5], device="cuda")

    def reference():
        lp, y_hat = math_ops.perturb_and_apply(lambda v: prior.log_prob(v.to(torch.float32)), y, u=noise,
                                               expected_grads=True)
        return y_hat, lp.sum(dim=(1, 2, 3)) / -float(np.log(2.0))

    def fused():
        return bottleneck_ops.factorized_bits(y, prior.base, 3, noise, expected_grads=True)

    outs = []
    for fn in (reference, fused):
        y.grad = None
        prior.zero_grad()
        y_hat, bits = fn()
        ((bits * w).sum() + (y_hat.float() ** 2).sum() * 1e-3).backward()
        outs.append((y_hat.detach(), bits.detach(), y.grad.detach().float().clone(),
                     [p.grad.detach().clone() for p in prior.parameters()]))
    (yr, br, gr, pr), (yf, bf, gf, pf) = outs
    assert torch.equal(yr, yf)
    assert torch.allclose(br, bf, rtol=1e-5, atol=1e-3)
    gtol = 2e-4 if dtype == torch.float32 else 2e-2
    assert torch.allclose(gr, gf, rtol=gtol, atol=gtol * gr.abs().max().item())
    for a, b in zip(pr, pf):
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-3 * max(a.abs().max().item(), 1e-3))
    # and through the entropy model's training call
    em = tfc.ContinuousBatchedEntropyModel(prior, coding_rank=3, compression=False, expected_grads=True)
    y2 = y.detach().clone().requires_grad_(True)
    _, bits2 = em(y2, training=True)
    bits2.sum().backward()
    assert torch.isfinite(y2.grad.float()).all()


@pytest.mark.parametrize("dtype,expected", [(torch.float32, False), (torch.float32, True), (torch.bfloat16, False)])
def test_fused_noisy_normal_bits_matches_torch_path(dtype, expected):
    """csrc/noisy_normal_bits.hip against the op-by-op evaluation (uniform_noise.py:117-156 over a Normal base,
    math_ops.py:157-216): same perturbed tensor, bits, gradients w.r.t. the input and the scale — including
    far tails (|y| up to 40 sigma), where the difference of cumulatives is taken in log space."""
    from compression_amd.distributions import uniform_noise
    from compression_amd.ops import bottleneck_ops, math_ops
    torch.manual_seed(21)
    shape = (3, 6, 5, 32)
    log_scale = torch.empty(shape, device="cuda").uniform_(-2.0, 3.0).requires_grad_(True)
    y = (torch.randn(shape, device="cuda") * torch.exp(log_scale.detach()) * 1.5).to(dtype)
    y.view(-1)[::97] *= 20.0                       # far tails
    y.requires_grad_(True)
    noise = (torch.rand(shape, device="cuda") - 0.5).to(dtype)
    w = torch.tensor([1.0, -0.5, 2.0], device="cuda")

    def reference():
        scale = torch.exp(log_scale)
        f = lambda v: uniform_noise.NoisyNormal(loc=0.0, scale=scale).log_prob(v.to(torch.float32))
        lp, y_hat = math_ops.perturb_and_apply(f, y, u=noise, expected_grads=expected)
        return y_hat, lp.sum(dim=(1, 2, 3)) / -float(np.log(2.0))

    def fused():
        return bottleneck_ops.noisy_normal_bits(y, torch.exp(log_scale), 3, noise, expected_grads=expected)

    outs = []
    for fn in (reference, fused):
        y.grad = None
        log_scale.grad = None
        y_hat, bits = fn()
        ((bits * w).sum() + (y_hat.float() ** 2).sum() * 1e-3).backward()
        outs.append((y_hat.detach(), bits.detach(), y.grad.detach().float().clone(), log_scale.grad.detach().clone()))
    (yr, br, gr, sr), (yf, bf, gf, sf) = outs
    assert torch.equal(yr, yf)
    assert torch.isfinite(bf).all() and torch.isfinite(gf).all() and torch.isfinite(sf).all()
    assert torch.allclose(br, bf, rtol=2e-5, atol=1e-2)
    gtol = 5e-4 if dtype == torch.float32 else 2e-2
    assert torch.allclose(gr, gf, rtol=gtol, atol=gtol * gr.abs().max().item())
    assert torch.allclose(sr, sf, rtol=5e-4, atol=5e-4 * sr.abs().max().item())


def test_location_scale_model_training_call_uses_fused_path():
    """LocationScaleIndexedEntropyModel(training=True) with NoisyNormal: bits equal the torch evaluation of
    the same perturbed tensor, gradients reach the bottleneck, the indexes (through scale_fn) and loc."""
    from compression_amd.distributions import uniform_noise
    torch.manual_seed(22)
    scale_fn = lambda i: torch.exp(-2.0 + 0.1 * i)
    em = tfc.LocationScaleIndexedEntropyModel(uniform_noise.NoisyNormal, 64, scale_fn, coding_rank=3)
    y = torch.randn(2, 5, 5, 16, device="cuda", requires_grad=True)
    idx = torch.empty(2, 5, 5, 16, device="cuda").uniform_(0, 63).requires_grad_(True)
    loc = torch.randn(2, 5, 5, 16, device="cuda", requires_grad=True)
    y_hat, bits = em(y, idx, loc=loc, training=True)
    assert y_hat.shape == y.shape and bits.shape == (2,)
    assert (y_hat - y).abs().max() <= 0.5 + 1e-6
    prior = uniform_noise.NoisyNormal(loc=0.0, scale=scale_fn(idx.detach()))
    want = prior.log_prob((y_hat - loc).detach()).sum(dim=(1, 2, 3)) / -float(np.log(2.0))
    assert torch.allclose(bits.detach(), want, rtol=1e-5, atol=1e-3)
    bits.sum().backward()
    for t in (y, idx, loc):
        assert t.grad is not None and torch.isfinite(t.grad).all() and t.grad.abs().sum() > 0


def _fused_normal_prob(x, loc, scale):
    """Per-element probabilities out of the fused NoisyNormal kernel (one coding unit per element)."""
    from compression_amd.ops import bottleneck_ops
    v = (torch.as_tensor(x, dtype=torch.float32, device="cuda") - loc).reshape(-1, 1)
    s = torch.full_like(v, float(scale))
    _, bits = bottleneck_ops.noisy_normal_bits(v, s, 1, None)
    return torch.exp2(-bits).cpu()


def test_fused_noisy_normal_closed_form_checks():
    """The checks of uniform_noise_test.py:46-61 (NoisyNormal) on the FUSED kernel: with the scale going to
    zero the density is a unit-width box, and on an integer grid it is a PMF."""
    mean = 10.0
    x = torch.linspace(mean - 1, mean + 1, 10)
    assert torch.allclose(_fused_normal_prob(x, mean, 1e-7), torch.tensor([0, 0, 0, 1, 1, 1, 1, 0, 0, 0.]), atol=1e-6)
    grid = 0.05 + torch.arange(-100, 100, dtype=torch.float32)
    assert abs(float(_fused_normal_prob(grid, 0.1, 0.3).sum()) - 1.0) < 1e-5
    # and against the closed form itself, in float64
    xs = torch.linspace(-6, 6, 49)
    want = (torch.special.ndtr((xs.double() - 0.1 + 0.5) / 0.7) - torch.special.ndtr((xs.double() - 0.1 - 0.5) / 0.7))
    assert torch.allclose(_fused_normal_prob(xs, 0.1, 0.7).double(), want, rtol=1e-5, atol=1e-9)


def test_fused_deep_factorized_closed_form_checks():
    """deep_factorized_test.py:119-124 (`test_uniform_is_special_case`) and the PMF property of
    uniform_noise_test.py:56-61 on the FUSED kernel (one channel, one coding unit per element)."""
    from compression_amd.ops import bottleneck_ops
    torch.manual_seed(0)

    def probs(prior, x):
        v = torch.as_tensor(x, dtype=torch.float32, device="cuda").reshape(-1, 1)
        _, bits = bottleneck_ops.factorized_bits(v, prior.base, 1, None)
        return torch.exp2(-bits.detach()).cpu()

    df = tfc.NoisyDeepFactorized(batch_shape=(1,), init_scale=1e-3).cuda()
    assert torch.allclose(probs(df, torch.linspace(-1, 1, 10)), torch.tensor([0, 0, 0, 1, 1, 1, 1, 0, 0, 0.]), atol=1e-5)
    df = tfc.NoisyDeepFactorized(batch_shape=(1,), init_scale=3.0).cuda()
    with torch.no_grad():
        for prm in df.parameters():
            prm.add_(0.2 * torch.randn_like(prm))
    grid = 0.3 + torch.arange(-200, 200, dtype=torch.float32)
    assert abs(float(probs(df, grid).sum()) - 1.0) < 1e-4


def _run_both(reference, fused, leaves, params):
    """(y_hat, bits, d/dleaves, d/dparams) of the two evaluations under the same scalar loss."""
    w = torch.tensor([1.0, -2.0, 0.5], device="cuda")
    outs = []
    for fn, prm in ((reference, params[0]), (fused, params[1])):
        for t in list(leaves) + list(prm):
            t.grad = None
        y_hat, bits = fn()
        ((bits * w).sum() + (y_hat.float() ** 2).sum() * 1e-3).backward()
        outs.append((y_hat.detach(), bits.detach().float(), [t.grad.detach().float().clone() for t in leaves],
                     [p.grad.detach().float().clone() for p in prm]))
    return outs


@pytest.mark.parametrize("expected", [False, True])
def test_fused_laplace_tail_deep_factorized(expected):
    """The *_tail entry points of csrc/factorized_bits.hip against the reference's `_log_prob`
    (continuous_base.py:298-334) evaluated op by op in FLOAT64 (a float32 evaluation of the mixture has no
    digits left where the tail matters), through math_ops.perturb_and_apply: bits, d/dy and the gradients of
    every prior parameter, with inputs on both sides of the 1e-10 switch and where the two components are of
    the same size."""
    from compression_amd.ops import bottleneck_ops, math_ops
    torch.manual_seed(31)
    C, m = 48, 1e-3
    prior = tfc.NoisyDeepFactorized(batch_shape=(C,), init_scale=0.5).cuda()
    with torch.no_grad():
        for prm in prior.parameters():
            prm.add_(0.2 * torch.randn_like(prm))
    prior64 = tfc.NoisyDeepFactorized(batch_shape=(C,), init_scale=0.5, dtype=torch.float64).cuda()
    prior64.load_state_dict(prior.state_dict())
    em = tfc.ContinuousBatchedEntropyModel(prior, coding_rank=3, laplace_tail_mass=m, expected_grads=expected)
    y = 3.0 * torch.randn(3, 5, 7, C, device="cuda")
    far = y.view(-1)[::53]
    far.copy_(torch.sign(far) * torch.empty_like(far).uniform_(4.0, 32.0))
    y.requires_grad_(True)
    noise = torch.rand(3, 5, 7, C, device="cuda") - 0.5

    with torch.no_grad():
        v = (y + noise).double()
        p_prior = prior64.prob(v)
        p_tail = m * tfc.NoisyLaplace(loc=0.0, scale=1.0, dtype=torch.float64).prob(v)
        mix = (1 - m) * p_prior + p_tail
        assert (mix < 1e-10).sum() >= 10                                    # the log m + log Q branch
        assert ((mix >= 1e-10) & (p_tail > 0.05 * p_prior)).sum() >= 10     # the tail is material, mixture branch

    def reference():
        lp, y_hat = math_ops.perturb_and_apply(lambda t: em._log_prob(prior64, t.double()), y, u=noise,
                                               expected_grads=expected)
        return y_hat, lp.sum(dim=(1, 2, 3)) / -float(np.log(2.0))

    def fused():
        return bottleneck_ops.factorized_bits(y, prior.base, 3, noise, expected_grads=expected, laplace_tail_mass=m)

    (yr, br, (gr,), pr), (yf, bf, (gf,), pf) = _run_both(
        reference, fused, [y], (list(prior64.parameters()), list(prior.parameters())))
    assert torch.equal(yr, yf)
    assert torch.isfinite(bf).all() and torch.isfinite(gf).all()
    assert torch.allclose(br, bf, rtol=1e-5, atol=1e-3)
    assert torch.allclose(gr, gf, rtol=2e-4, atol=2e-4 * gr.abs().max().item())
    for a, b in zip(pr, pf):
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-3 * max(a.abs().max().item(), 1e-3))
    # without the tail the result differs (the far inputs cost more bits): the option is not a no-op
    _, plain = bottleneck_ops.factorized_bits(y.detach(), prior.base, 3, noise)
    assert (plain - bf > 1.0).all()
    # the entropy model's training call takes the fused kernels
    calls = []
    orig = bottleneck_ops.factorized_bits
    try:
        bottleneck_ops.factorized_bits = lambda *a, **k: (calls.append(k), orig(*a, **k))[1]
        y2 = y.detach().clone().requires_grad_(True)
        y_hat2, bits2 = em(y2, training=True)
    finally:
        bottleneck_ops.factorized_bits = orig
    assert len(calls) == 1 and calls[0]["laplace_tail_mass"] == m
    want = em._log_prob(prior64, y_hat2.detach().double()).sum(dim=(1, 2, 3)) / -float(np.log(2.0))
    assert torch.allclose(bits2.detach(), want.float(), rtol=1e-5, atol=1e-3)
    bits2.sum().backward()
    assert torch.isfinite(y2.grad).all()


@pytest.mark.parametrize("dtype,expected", [(torch.float32, False), (torch.float32, True), (torch.bfloat16, False)])
def test_fused_laplace_tail_noisy_normal(dtype, expected):
    """The *_tail entry points of csrc/noisy_normal_bits.hip against `_log_prob` (continuous_base.py:298-334)
    in float64 over a NoisyNormal(0, scale): bits, d/dy, d/dscale, inputs out to where the normal's mass
    underflows and only the log m + log Q branch is left."""
    from compression_amd.distributions import uniform_noise
    from compression_amd.ops import bottleneck_ops, math_ops
    torch.manual_seed(32)
    m = 1e-3
    shape = (3, 6, 5, 32)
    log_scale = torch.empty(shape, device="cuda").uniform_(-2.0, 1.5).requires_grad_(True)
    y = (torch.randn(shape, device="cuda") * torch.exp(log_scale.detach()) * 1.5).to(dtype)
    y.view(-1)[::37] *= 12.0
    y.requires_grad_(True)
    noise = (torch.rand(shape, device="cuda") - 0.5).to(dtype)
    em = tfc.LocationScaleIndexedEntropyModel(uniform_noise.NoisyNormal, 64, lambda i: torch.exp(-2.0 + 0.1 * i),
                                              coding_rank=3, laplace_tail_mass=m, expected_grads=expected,
                                              bottleneck_dtype=dtype)

    def reference():
        prior = uniform_noise.NoisyNormal(loc=0.0, scale=torch.exp(log_scale).double(), dtype=torch.float64)
        lp, y_hat = math_ops.perturb_and_apply(lambda t: em._log_prob(prior, t.double()), y, u=noise,
                                               expected_grads=expected)
        return y_hat, lp.sum(dim=(1, 2, 3)) / -float(np.log(2.0))

    def fused():
        return bottleneck_ops.noisy_normal_bits(y, torch.exp(log_scale), 3, noise, expected_grads=expected,
                                                laplace_tail_mass=m)

    with torch.no_grad():
        v = (y + noise).double()
        prior = uniform_noise.NoisyNormal(loc=0.0, scale=torch.exp(log_scale).double(), dtype=torch.float64)
        p_tail = m * uniform_noise.NoisyLaplace(loc=0.0, scale=1.0, dtype=torch.float64).prob(v)
        mix = (1 - m) * prior.prob(v) + p_tail
        assert (mix < 1e-10).sum() >= 10 and ((mix >= 1e-10) & (p_tail > 0.05 * prior.prob(v))).sum() >= 10

    (yr, br, (gr, sr), _), (yf, bf, (gf, sf), _) = _run_both(reference, fused, [y, log_scale], ([], []))
    assert torch.equal(yr, yf)
    assert torch.isfinite(bf).all() and torch.isfinite(gf).all() and torch.isfinite(sf).all()
    assert torch.allclose(br, bf, rtol=2e-5, atol=1e-2)
    gtol = 5e-4 if dtype == torch.float32 else 2e-2
    assert torch.allclose(gr, gf, rtol=gtol, atol=gtol * gr.abs().max().item())
    assert torch.allclose(sr, sf, rtol=5e-4, atol=5e-4 * sr.abs().max().item())
    # the model's training call (loc given: the model shifts the bottleneck, the prior's own loc is 0)
    calls = []
    orig = bottleneck_ops.noisy_normal_bits
    try:
        bottleneck_ops.noisy_normal_bits = lambda *a, **k: (calls.append(k), orig(*a, **k))[1]
        idx = torch.empty(shape, device="cuda").uniform_(0, 63)
        loc = torch.randn(shape, device="cuda").to(dtype)
        y_hat2, bits2 = em(y.detach(), idx, loc=loc, training=True)
    finally:
        bottleneck_ops.noisy_normal_bits = orig
    assert len(calls) == 1 and calls[0]["laplace_tail_mass"] == m
    assert torch.isfinite(bits2).all() and (bits2 > 0).all()
    if dtype == torch.float32:          # (in bfloat16, y_hat - loc is not the value the kernel saw)
        prior2 = uniform_noise.NoisyNormal(loc=0.0, scale=torch.exp(-2.0 + 0.1 * idx).double(), dtype=torch.float64)
        want = em._log_prob(prior2, (y_hat2 - loc).double()).sum(dim=(1, 2, 3)) / -float(np.log(2.0))
        assert torch.allclose(bits2.float(), want.float(), rtol=1e-4, atol=1e-2)


def test_indexed_model_with_a_prior_location_keeps_the_tail_unfused():
    """A NoisyNormal whose location comes from the indexes: the Laplace component of the tail sits at 0 of the
    unshifted bottleneck (continuous_base.py:298-334), which the shifted kernel cannot see — that
    configuration must stay on the op-by-op path, and agree with `_log_prob`."""
    from compression_amd.distributions import uniform_noise
    from compression_amd.ops import bottleneck_ops
    torch.manual_seed(33)
    em = tfc.ContinuousIndexedEntropyModel(
        uniform_noise.NoisyNormal, (8,), dict(loc=lambda i: 0.5 * i - 2.0, scale=lambda i: torch.exp(0.2 * i - 1.0)),
        coding_rank=1, channel_axis=None, laplace_tail_mass=1e-3)
    y = 4.0 * torch.randn(5, 40, device="cuda")
    idx = torch.randint(0, 8, (5, 40), device="cuda").float()
    calls = []
    orig = bottleneck_ops.noisy_normal_bits
    try:
        bottleneck_ops.noisy_normal_bits = lambda *a, **k: (calls.append(k), orig(*a, **k))[1]
        y_hat, bits = em(y, idx, training=True)
    finally:
        bottleneck_ops.noisy_normal_bits = orig
    assert not calls
    want = em._log_prob(em._make_prior(idx), y_hat).sum(dim=1) / -float(np.log(2.0))
    assert torch.allclose(bits, want, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("dist_cls", ["NoisyNormalMixture", "NoisyLogisticMixture"])
def test_indexed_model_with_a_mixture_prior_round_trip(dist_cls):
    """A noisy mixture prior (uniform_noise.py:203-319) through table building (tails located on the
    noise-free mixture by the iterative solver, helpers.py:29-104) and the indexed coder: exact round trip,
    coded size close to the model's own estimate."""
    torch.manual_seed(41)
    two = torch.tensor([-2.0, 2.0])
    em = tfc.ContinuousIndexedEntropyModel(
        getattr(tfc, dist_cls), index_ranges=(6, 4), channel_axis=-1, coding_rank=1, compression=True,
        parameter_fns=dict(loc=lambda i: i[..., 0:1] - 3 + two.to(i.device),
                           scale=lambda i: torch.exp(0.3 * i[..., 1:2]) * torch.ones(2, device=i.device),
                           weight=lambda i: torch.tensor([0.4, 0.6], device=i.device).expand(i.shape[:-1] + (2,))))
    idx = torch.stack([torch.randint(0, 6, (4, 2000)), torch.randint(0, 4, (4, 2000))], dim=-1).float()
    pick = torch.where(torch.rand(4, 2000) < 0.4, -2.0, 2.0)
    y = idx[..., 0] - 3 + pick + torch.randn(4, 2000) * torch.exp(0.3 * idx[..., 1])
    strings = em.compress(y, idx)
    assert strings.shape == (4,)
    assert torch.equal(em.decompress(strings, idx).cpu(), torch.round(y))
    _, bits = em(y.cuda(), idx.cuda(), training=False)
    coded = np.array([8 * len(bytes(s)) for s in strings], dtype=np.float64)
    assert np.all(coded < bits.cpu().numpy() * 1.02 + 64) and np.all(coded > bits.cpu().numpy() * 0.95)
