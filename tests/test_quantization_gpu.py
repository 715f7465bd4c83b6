"""GPU tier: StochasticRound (HIP kernel through the op API -> C ABI) against the outputs of the
reference kernel file (tests/golden/stochastic_round.npz), against the CPU oracle on large tensors, and
the properties the reference's own test checks (python/ops/quantization_ops_test.py:24-75).
Integer outputs: every comparison with golden / oracle is equality."""
import time

import numpy as np
import pytest
import torch

from oracle.make_golden import STOCHASTIC_CASES, STOCHASTIC_N, stochastic_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tfc():
    import compression_amd
    return compression_amd


def to_device(inputs):
    if isinstance(inputs, tuple):
        bits, code = inputs
        t = torch.from_numpy(bits.view(np.int16).copy()).cuda()
        return t.view(torch.bfloat16 if code == 1 else torch.float16)
    return torch.from_numpy(np.ascontiguousarray(inputs)).cuda()


@pytest.mark.parametrize("name", sorted(STOCHASTIC_CASES))
def test_golden(tfc, golden, name):
    code, step, seed = STOCHASTIC_CASES[name]
    got = tfc.stochastic_round(to_device(stochastic_inputs(STOCHASTIC_N, code)), step, seed)
    assert got.dtype == torch.int32
    want = golden("stochastic_round.npz")[name].astype(np.int32)
    assert (got.cpu().numpy() == want).all()


@pytest.mark.parametrize("n", [1, 63, 255, 256, 257, 16384, 16385, 5 * 16384 + 300, (1 << 26) + 12345])
def test_matches_oracle(tfc, port, n):
    """Sizes around the lane (256) and wave (16384) segment boundaries; the largest uses the jump
    matrices up to T^(2^26)."""
    rng = np.random.default_rng(n)
    x = rng.uniform(-1000, 1000, n).astype(np.float32)
    seed = (n & 0xFFFF, 17, -3)
    got = tfc.stochastic_round(torch.from_numpy(x).cuda(), 0.37, seed).cpu().numpy()
    want = port.stochastic_round(x, 0.37, seed)
    assert (got == want).all()


def test_shape_seed_shape_and_half_types(tfc, port):
    rng = np.random.default_rng(5)
    x = rng.uniform(-50, 50, (3, 70, 129)).astype(np.float32)
    seed = np.array([[1, 2], [3, 4]], np.int32)
    for dtype, code in ((torch.bfloat16, 1), (torch.float16, 2)):
        t = torch.from_numpy(x).cuda().to(dtype)
        bits = t.view(torch.int16).cpu().numpy().view(np.uint16)
        got = tfc.stochastic_round(t, torch.tensor(0.5), torch.from_numpy(seed))
        assert got.shape == t.shape
        assert (got.cpu().numpy() == port.stochastic_round((bits, code), 0.5, seed)).all()
    with pytest.raises(ValueError, match="step_size must be a scalar"):
        tfc.stochastic_round(torch.zeros(4).cuda(), torch.ones(2), (1,))
    with pytest.raises(TypeError):
        tfc.stochastic_round(torch.zeros(4, dtype=torch.float64).cuda(), 1.0, (1,))
    assert tfc.stochastic_round(torch.zeros(0).cuda(), 1.0, (1,)).shape == (0,)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_difference_is_at_most_one(tfc, dtype):
    values = (torch.rand(100, device="cuda") * 200 - 100).to(dtype)
    rounded = tfc.stochastic_round(values, 1.0, ())
    assert rounded.dtype == torch.int32
    assert (values.float() - rounded.float()).abs().max() <= 1


def test_seeds(tfc):
    values = torch.rand(100, device="cuda") * 200 - 100
    r1 = tfc.stochastic_round(values, 1.0, (123, 456))
    r2 = tfc.stochastic_round(values, 1.0, (123, 456))
    r3 = tfc.stochastic_round(values, 1.0, (456, 789))
    assert torch.equal(r1, r2) and not torch.equal(r1, r3)
    c1 = tfc.stochastic_round(values, 1.0, ())
    time.sleep(0.01)
    c2 = tfc.stochastic_round(values, 1.0, ())
    assert not torch.equal(c1, c2)


@pytest.mark.parametrize("step_size", [1.0, 0.75, 1e-4])
def test_integers_and_half_integers(tfc, step_size):
    ints = torch.randint(-100, 100, (100,), device="cuda", dtype=torch.int32)
    rounded = tfc.stochastic_round(ints.float() * step_size, step_size, ())
    assert torch.equal(rounded, ints)
    halves = torch.arange(-10, 10, device="cuda", dtype=torch.float32) + 0.5
    rounded = tfc.stochastic_round(halves * step_size, step_size, ())
    assert (halves - rounded.float()).abs().max() <= 0.5


def test_rounding_is_unbiased(tfc):
    values = torch.rand(20, device="cuda") * 200 - 100
    rounded = tfc.stochastic_round(values.expand(100000, 20).contiguous(), 1.0, ())
    assert (rounded.float().mean(0) - values).abs().max() < 1e-2
