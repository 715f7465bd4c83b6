"""CPU tier: python/ops/math_ops.py — the reference's math_ops_test.py cases (bounds with their three gradient
rules, perturb_and_apply)."""
import pytest
import scipy.stats
import torch

import compression_amd as tfc
from compression_amd.ops import math_ops


@pytest.mark.parametrize("gradient", ["disconnected", "identity", "identity_if_towards"])
@pytest.mark.parametrize("which", ["upper", "lower"])
def test_bounds_have_correct_outputs_and_gradients(which, gradient):
    fn = math_ops.upper_bound if which == "upper" else math_ops.lower_bound
    grads = {}
    for name, seed in (("p", 1.0), ("n", -1.0)):
        inputs = torch.tensor([-1.0, 1.0], requires_grad=True)
        outputs = fn(inputs, 0, gradient=gradient)
        grads[name], = torch.autograd.grad(outputs, inputs, seed * torch.ones(2))
    assert outputs.tolist() == ([-1, 0] if which == "upper" else [0, 1])
    want = {("upper", "disconnected"): ([1, 0], [-1, 0]), ("upper", "identity"): ([1, 1], [-1, -1]),
            ("upper", "identity_if_towards"): ([1, 1], [-1, 0]),
            ("lower", "disconnected"): ([0, 1], [0, -1]), ("lower", "identity"): ([1, 1], [-1, -1]),
            ("lower", "identity_if_towards"): ([0, 1], [-1, -1])}[(which, gradient)]
    assert grads["p"].tolist() == want[0] and grads["n"].tolist() == want[1]
    with pytest.raises(ValueError):
        fn(torch.zeros(1, 2), 0, gradient="invalid")


def test_perturb_and_apply_noise():
    torch.manual_seed(0)
    x = torch.randn(10000)
    y, x_plus_u = math_ops.perturb_and_apply(lambda t: t.clone(), x, expected_grads=True)
    u0, u1 = x_plus_u - x, y - x
    assert torch.allclose(u0, u1, atol=1e-6) and (u0.abs() <= 0.5).all()
    _, p = scipy.stats.kstest(u0.numpy(), "uniform", (-0.5, 1.0))
    assert p > 1e-6


def test_perturb_and_apply_expected_gradients():
    # soft_round: the expected slope over a unit of noise is exactly 1
    x = torch.linspace(-2.0, 2.0, 200, requires_grad=True)
    y = math_ops.perturb_and_apply(tfc.soft_round, x, 7.0, expected_grads=True)[0]
    dx, = torch.autograd.grad(y.sum(), x)
    assert torch.allclose(dx, torch.ones_like(dx), atol=1e-5)
    # a parabola: f(x + .5) - f(x - .5)
    f = lambda t, a: a * t * t
    x = torch.linspace(-2.0, 2.0, 200, requires_grad=True)
    y = math_ops.perturb_and_apply(f, x, 7.0, expected_grads=True)[0]
    dx, = torch.autograd.grad(y.sum(), x)
    assert torch.allclose(dx, (f(x + 0.5, 7.0) - f(x - 0.5, 7.0)).detach(), atol=1e-4)
    with pytest.raises(ValueError):
        math_ops.perturb_and_apply(f, x, 7.0, u=torch.zeros(200), x_plus_u=x)
