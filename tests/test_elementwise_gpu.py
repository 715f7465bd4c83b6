"""GPU tier: the fused elementwise passes of the model pipelines (csrc/elementwise.hip) against the torch expressions
they replace (models/bls2017.py:164-190, bmshj2018.py:219-264; continuous_indexed.py:272-296)."""
import numpy as np
import pytest
import torch

from compression_amd.layers import functional

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_image_to_unit(dtype):
    x = torch.arange(256, dtype=torch.uint8).repeat(41)[:10001].reshape(1, 73, 137, 1).contiguous().cuda()
    got = functional.image_to_unit(x, dtype)
    # IEEE division, as the reference's tf.cast(x, dtype) / 255 (torch's device kernel multiplies by the reciprocal:
    # one ulp off for some float32 values — the host's numpy division is the yardstick)
    want = torch.from_numpy(x.cpu().numpy().astype(np.float32) / np.float32(255.0)).to(dtype).cuda()
    assert got.dtype == dtype and torch.equal(got, want)
    assert (got.float() - (x.to(dtype) / 255.0).float()).abs().max() <= 2.0 ** -23
    # a contiguous tensor whose first element is not 16-byte aligned (the kernel's element-by-element path)
    xo = x.view(-1)[3:]
    assert torch.equal(functional.image_to_unit(xo, dtype), want.view(-1)[3:])
    # non-contiguous input: the torch expression
    xt = x.expand(2, 73, 137, 3)[:, ::2]
    assert torch.equal(functional.image_to_unit(xt, dtype), xt.to(dtype) / 255.0)     # (the fallback IS that expression)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_unit_to_image(dtype):
    torch.manual_seed(0)
    # values around every rounding boundary k + 1/2, below 0 and above 1
    k = torch.arange(-8, 264, dtype=torch.float32)
    cases = torch.cat([(k + d) / 255.0 for d in (-0.5, -0.25, 0.0, 0.25, 0.4999, 0.5, 0.5001)])
    x = torch.cat([cases, torch.rand(50020) * 1.2 - 0.1]).to(dtype).cuda()
    got = functional.unit_to_image(x)
    want = torch.clamp(torch.round((x * 255.0).float()), 0, 255).to(torch.uint8)     # the product in x's dtype
    assert got.dtype == torch.uint8 and torch.equal(got, want)
    assert torch.equal(functional.unit_to_image(x[5:]), want[5:])          # (base not 16-byte aligned)
    xs = x.reshape(-1, 3)[:, :2]
    assert torch.equal(functional.unit_to_image(xs), torch.clamp(torch.round((xs * 255.0).float()), 0, 255).to(torch.uint8))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_index_prepare_equals_the_bound_ops(dtype):
    from compression_amd.ops import math_ops
    torch.manual_seed(1)
    idx = (torch.randn(3, 17, 23, 5) * 40 + 20).to(dtype).cuda()
    idx.view(-1)[:6] = torch.tensor([-0.0, 0.49, 62.999, 63.0, 63.5, 1e9]).to(dtype)
    got = functional.index_prepare(idx, 64)
    want = math_ops.upper_bound(math_ops.lower_bound(idx, 0), 63).to(torch.int32)
    assert got.dtype == torch.int32 and torch.equal(got, want) and int(got.min()) >= 0 and int(got.max()) <= 63
    got1 = functional.index_prepare(idx.view(-1)[1:], 64)                     # (base not 16-byte aligned)
    assert got1 is None or torch.equal(got1, want.view(-1)[1:])
    assert functional.index_prepare(idx.to(torch.int32), 64) is None          # integer indexes: the torch ops
    assert functional.index_prepare(idx[..., :3], 64) is None                 # not contiguous
