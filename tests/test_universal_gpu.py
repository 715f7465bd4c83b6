"""GPU tier: universal-quantisation entropy models on the HIP coder (entropy_models/universal.py;
the reference's universal_test.py checks: round trip under the shared dither, rate estimate vs coded size)."""
import numpy as np
import pytest
import torch

import compression_amd as tfc
from compression_amd.entropy_models import universal

pytestmark = pytest.mark.gpu


def test_batched_roundtrip_and_rate():
    torch.manual_seed(1)
    C = 16
    prior = tfc.NoisyNormal(loc=torch.linspace(-1, 1, C), scale=torch.linspace(0.5, 6.0, C))
    em = tfc.UniversalBatchedEntropyModel(prior, coding_rank=3, compression=True, num_noise_levels=15)
    assert em.cdf_offset.shape[0] == 15 * C                    # one table per (offset level, channel)
    y = (torch.randn(3, 9, 11, C) * torch.linspace(0.5, 6.0, C) + torch.linspace(-1, 1, C)).cuda()
    strings = em.compress(y)
    assert strings.shape == (3,)
    y_hat = em.decompress(strings, (9, 11))
    assert y_hat.shape == y.shape
    # what the decoder must see: round(y - o) + o with the shared offsets
    _, offset = em._compute_indexes_and_offset((9, 11))
    want = torch.round(y - offset.cuda()) + offset.cuda()
    assert torch.equal(y_hat, want)
    assert (y_hat - y).abs().max() <= 0.5 + 1e-6
    # the coded size is the evaluation-mode estimate up to the coder's overhead
    _, bits = em(y, training=False)
    coded = np.array([8 * len(bytes(s)) for s in strings], dtype=np.float64)
    assert np.all(coded >= bits.cpu().numpy() * 0.98) and np.all(coded <= bits.cpu().numpy() * 1.05 + 64)
    # a damaged string fails the sanity check or decodes to something else, never silently the same
    with pytest.raises(ValueError):
        em.compress(torch.zeros(0, 9, 11, C))


def test_batched_offsets_differ_from_plain_rounding():
    """The dither is actually applied: with a wide prior the reconstruction error is uniform in
    (-1/2, 1/2) and NOT concentrated on the integer grid as with round()."""
    torch.manual_seed(2)
    prior = tfc.NoisyNormal(loc=0.0, scale=torch.full((4,), 8.0))
    em = tfc.UniversalBatchedEntropyModel(prior, coding_rank=2, compression=True)
    y = (torch.randn(2, 4000, 4) * 8).cuda()
    y_hat = em.decompress(em.compress(y), (4000,))
    frac = (y_hat - torch.floor(y_hat)).cpu().numpy().ravel()
    levels = np.unique(np.round(frac * 16).astype(int) % 16)
    assert len(levels) == 15 and 8 not in levels            # offsets (k + 1)/16 - 1/2, k = 0..14: never 0 mod 1


def test_indexed_roundtrip_and_rate():
    torch.manual_seed(3)
    em = tfc.UniversalIndexedEntropyModel(
        tfc.NoisyNormal, index_ranges=(32,), parameter_fns=dict(loc=lambda i: 0.0, scale=lambda i: torch.exp(-1.0 + 0.12 * i[..., 0])),
        coding_rank=2, compression=True, num_noise_levels=11)
    assert em.index_ranges == (11, 32) and em.cdf_offset.shape[0] == 11 * 32
    idx = torch.randint(0, 32, (4, 300, 5, 1)).float().cuda()
    scale = torch.exp(-1.0 + 0.12 * idx[..., 0])
    y = torch.randn(4, 300, 5, device="cuda") * scale
    strings = em.compress(y, idx)
    assert strings.shape == (4,)
    y_hat = em.decompress(strings, idx)
    off = universal._offset_indexes_to_offset(universal.stateless_offset_indexes((4, 300, 5), 11), 11, torch.float32).cuda()
    assert torch.equal(y_hat, torch.round(y - off) + off)
    _, bits = em(y, idx, training=False)
    coded = np.array([8 * len(bytes(s)) for s in strings], dtype=np.float64)
    assert np.all(coded >= bits.cpu().numpy() * 0.97) and np.all(coded <= bits.cpu().numpy() * 1.06 + 64)
    # training call: gradients reach the bottleneck and the indexes
    yg = y.clone().requires_grad_(True)
    ig = idx.clone().requires_grad_(True)
    _, bt = em(yg, ig, training=True)
    bt.sum().backward()
    assert torch.isfinite(yg.grad).all() and torch.isfinite(ig.grad).all() and ig.grad.abs().sum() > 0


def test_batched_roundtrip_with_a_soft_rounded_prior():
    """The universal models' usual prior (round_adapters.py:253-290: soft-rounded normal + uniform noise): table
    building through the monotonic adapter's tails, exact round trip, coded size near the estimate."""
    torch.manual_seed(4)
    C = 8
    prior = tfc.NoisySoftRoundedNormal(loc=torch.linspace(-1, 1, C), scale=torch.linspace(0.7, 5.0, C), alpha=3.0)
    em = tfc.UniversalBatchedEntropyModel(prior, coding_rank=2, compression=True, num_noise_levels=15)
    y = tfc.soft_round(torch.randn(3, 1500, C) * torch.linspace(0.7, 5.0, C) + torch.linspace(-1, 1, C), 3.0).cuda()
    strings = em.compress(y)
    y_hat = em.decompress(strings, (1500,))
    _, offset = em._compute_indexes_and_offset((1500,))
    assert torch.equal(y_hat, torch.round(y - offset.cuda()) + offset.cuda())
    _, bits = em(y, training=False)
    coded = np.array([8 * len(bytes(s)) for s in strings], dtype=np.float64)
    assert np.all(coded >= bits.cpu().numpy() * 0.97) and np.all(coded <= bits.cpu().numpy() * 1.06 + 64)
