"""GPU tier, slow: BASELINE configs 1 and 4 at their FULL batch (bls2017: 512 x 256x256, bmshj2018: 128 x 768x512,
calibrated hyperprior) — every image's main-latent string must equal what the reference coder (oracle/_ref when
present, else its restatement) produces on the very symbols the model's coder read, and decompress must return the
quantised latents.  bench.py prints the same comparison in its `models` objects; this is the copy pytest carries
(models/bls2017.py:164-190, models/bmshj2018.py:219-264)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = [pytest.mark.gpu, pytest.mark.slow]


@pytest.mark.parametrize("workload,group", [("bls2017", 1), ("bmshj2018", 1), ("bmshj2018", 4)])
def test_full_batch_strings_equal_the_reference_coder(workload, group):
    import bench
    import compression_amd as tfc
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    model, x, batch, hw, hist = bench.make_model(workload, torch.bfloat16, device, 0)
    assert batch == (512 if workload == "bls2017" else 128)
    if group == 1:
        out = model.compress(x, device_result=True)
        x_hat, oks = model.decompress(*out, defer_sanity=True)
    else:
        # several batches behind one coder launch per stage (the pipelined lane kernels); the first is checked
        packed = model.compress_many([torch.roll(x, k, 0) for k in range(group)])
        x_hats, oks = model.decompress_many(packed)
        out, x_hat = packed[0], x_hats[0]
    assert all(bool(ok.cpu().all()) for ok in oks)
    assert x_hat.shape == x.shape
    strings = tfc.fetch_strings(out[0])
    y_coded = out[0].coder_inputs[0]
    want = model.entropy_model.quantize(y_coded) if workload == "bls2017" else torch.round(y_coded.float()).to(y_coded.dtype)
    assert torch.equal(x_hat._tfc_keep[-1], want), "decompress did not return the quantised latents"
    res = bench.model_cpu_baseline(model, out[0], strings, hw)
    assert res["images_compared"] == batch and res["images_differing"] == 0 and res["bytes_identical_to_gpu"]
    if hist is not None:
        assert (hist > 0).sum() >= 40      # the calibrated index field really spreads over the scale tables
