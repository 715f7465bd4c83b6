"""GPU tier: PmfToQuantizedCdf kernel — equality with tables produced by the reference's own kernel
file (tests/golden/pmf_to_cdf.npz, tie-heavy: symmetric, flat and two-level tables), the reference's
invariant tests (pmf_to_cdf_kernels_test.cc:70-143) and equality with the CPU oracle on random and on
tied inputs."""
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


def test_golden_tables_with_ties(golden):
    import compression_amd as tfc
    g = golden("pmf_to_cdf.npz")
    for k in range(int(g["count"])):
        pmf, prec = g[f"pmf{k}"], int(g[f"precision{k}"])
        cdf = tfc.pmf_to_quantized_cdf(torch.from_numpy(pmf).cuda(), prec).cpu().numpy()
        assert (cdf == g[f"cdf{k}"]).all(), (k, pmf.shape, prec)


def test_tied_inputs_match_oracle(port):
    """Few distinct masses => long runs of equal penalties, rows up to 1500 wide (the quicksort part of
    the order runs many levels deep)."""
    import compression_amd as tfc
    rng = np.random.default_rng(11)
    for n in (3, 40, 333, 1500):
        for scale in (0.4, 1.0, 1.6):
            pmf = rng.integers(0, 5, (5, n)).astype(np.float32) + np.float32(0.01)
            pmf = pmf / pmf.sum(-1, keepdims=True) * np.float32(scale)
            cdf = tfc.pmf_to_quantized_cdf(torch.from_numpy(pmf).cuda(), 12).cpu().numpy()
            assert (cdf == port.pmf_to_quantized_cdf(pmf, 12)).all(), (n, scale)


def test_validation():
    import compression_amd as tfc
    with pytest.raises(ValueError, match="non-finite or negative"):
        tfc.pmf_to_quantized_cdf(torch.tensor([[0.5, float("nan")]]).cuda(), 8)
    with pytest.raises(ValueError, match="at least 2"):
        tfc.pmf_to_quantized_cdf(torch.tensor([[0.5]]).cuda(), 8)
    with pytest.raises(ValueError, match=r"precision` must be in \[1, 16\]"):
        tfc.pmf_to_quantized_cdf(torch.tensor([[0.5, 0.5]]).cuda(), 17)
