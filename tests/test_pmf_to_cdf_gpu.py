"""GPU tier: PmfToQuantizedCdf kernel — the reference's own invariant tests
(pmf_to_cdf_kernels_test.cc:70-143) plus equality with the CPU oracle on
tie-free inputs (ties are platform-specific in the reference, see
compression_amd/csrc/pmf_to_cdf.hip)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_invariants_and_oracle_equality(port):
    import compression_amd as tfc
    rng = np.random.Generator(np.random.PCG64(4))
    for prec in (4, 10, 12, 16):
        for n in (2, 17, 64, 65, 300, 1500):
            if n > (1 << prec):
                continue
            for scale in (0.5, 1.0, 1.7):
                w = rng.random((6, n)) ** 2 + 1e-4
                pmf = (w / w.sum(-1, keepdims=True) * scale).astype(np.float32)
                cdf = tfc.pmf_to_quantized_cdf(torch.from_numpy(pmf).cuda(), prec).cpu().numpy()
                assert cdf.shape == (6, n + 1)
                assert (cdf[:, 0] == 0).all() and (cdf[:, -1] == 1 << prec).all()
                assert (np.diff(cdf, axis=-1) >= 1).all()
                want = port.pmf_to_quantized_cdf(pmf, prec)
                # random continuous masses have no exact ties
                assert (cdf == want).all(), (prec, n, scale)


def test_validation():
    import compression_amd as tfc
    with pytest.raises(ValueError, match="non-finite or negative"):
        tfc.pmf_to_quantized_cdf(torch.tensor([[0.5, float("nan")]]).cuda(), 8)
    with pytest.raises(ValueError, match="at least 2"):
        tfc.pmf_to_quantized_cdf(torch.tensor([[0.5]]).cuda(), 8)
    with pytest.raises(ValueError, match=r"precision` must be in \[1, 16\]"):
        tfc.pmf_to_quantized_cdf(torch.tensor([[0.5, 0.5]]).cuda(), 17)
