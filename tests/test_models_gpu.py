"""GPU tier: end-to-end compress/decompress of the two target models (BASELINE
configs 1 and 4 at reduced batch): transforms + entropy coding through the HIP path,
byte parity of the coded strings against the CPU oracle given the same latents."""
import numpy as np
import pytest
import torch

import compression_amd as tfc
from compression_amd import synthetic

pytestmark = pytest.mark.gpu


def test_bls2017_single_image_plumbing(port):
    """Config 1: one 256x256x3 image through analysis -> coder -> synthesis."""
    torch.manual_seed(0)
    model = tfc.models.BLS2017Model(num_filters=192).cuda().init_compression()
    x = torch.from_numpy(synthetic.lowpass_images(1, 256, 256))[0].cuda()
    string, x_shape, y_shape = model.compress(x)
    assert string.shape == (1,) and x_shape == (256, 256) and y_shape == (16, 16)
    x_hat = model.decompress(string, x_shape, y_shape)
    assert x_hat.shape == (1, 256, 256, 3) and x_hat.dtype == torch.uint8
    # the coded string is exactly what the oracle produces for the same latents and tables
    em = model.entropy_model
    y = model.analysis_transform(x[None].float())
    off = em.quantization_offset
    sym = torch.round(y - off.cuda() if off is not None else y).to(torch.int32).cpu()
    sym = sym.reshape(1, -1) - em.cdf_offset.repeat(16 * 16)
    want, _, _ = port.encode(em.cdf.numpy(), sym.numpy())
    assert bytes(string[0]) == want[0]
    # decode side: decompress(compress(x)) equals synthesis(quantize(y))
    ref = model.synthesis_transform(em.quantize(y))
    ref = torch.clamp(torch.round(ref), 0, 255).to(torch.uint8)
    assert torch.equal(x_hat, ref)


def test_stored_tables_are_loaded_not_regenerated():
    """A checkpoint saved after init_compression() carries the tables; the receiving side loads them
    (continuous_base.py:175-184) — here the prior is changed AFTER the tables were fixed, so regenerated
    tables would differ and decode garbage."""
    from compression_amd.models import codec_io
    torch.manual_seed(5)
    sender = tfc.models.BLS2017Model(num_filters=64).cuda().init_compression()
    with torch.no_grad():
        for p in sender.prior.parameters():      # the prior drifts after the tables were built
            p.add_(0.05 * torch.randn_like(p))
    sd = {k: v.cpu() for k, v in sender.state_dict().items()}
    assert any(k.endswith("_cdf") for k in sd)
    receiver = codec_io.load_checkpoint(tfc.models.BLS2017Model(num_filters=64).cuda(), sd)
    assert torch.equal(receiver.entropy_model.cdf.cpu(), sender.entropy_model.cdf.cpu())
    x = torch.from_numpy(synthetic.lowpass_images(2, 96, 64)).cuda()
    strings, x_shape, y_shape = sender.compress(x)
    assert torch.equal(receiver.decompress(strings, x_shape, y_shape), sender.decompress(strings, x_shape, y_shape))
    # a checkpoint without tables builds them from its prior
    plain = {k: v for k, v in sd.items() if not k.startswith("entropy_model.")}
    fresh = codec_io.load_checkpoint(tfc.models.BLS2017Model(num_filters=64).cuda(), plain)
    assert fresh.entropy_model.cdf.numel() > 0


def test_bls2017_batch_and_odd_sizes():
    torch.manual_seed(1)
    model = tfc.models.BLS2017Model(num_filters=64).cuda().init_compression()
    x = torch.from_numpy(synthetic.lowpass_images(3, 100, 77)).cuda()
    strings, x_shape, y_shape = model.compress(x)
    assert strings.shape == (3,) and y_shape == (7, 5)
    x_hat = model.decompress(strings, x_shape, y_shape)
    assert x_hat.shape == (3, 100, 77, 3)
    one, _, _ = model.compress(x[1])
    assert bytes(one[0]) == bytes(strings[1])      # batch element == single-image call


def test_bmshj2018_roundtrip_kodak_shape():
    """Config 4 at batch 2: 768x512 images through the hyperprior model."""
    torch.manual_seed(2)
    model = tfc.models.BMSHJ2018Model(num_filters=192).cuda().init_compression()
    x = torch.from_numpy(synthetic.lowpass_images(2, 512, 768)).cuda()
    string, side_string, x_shape, y_shape, z_shape = model.compress(x)
    assert string.shape == side_string.shape == (2,)
    assert y_shape == (32, 48) and z_shape == (8, 12)
    x_hat = model.decompress(string, side_string, x_shape, y_shape, z_shape)
    assert x_hat.shape == (2, 512, 768, 3)
    # determinism: the decoder reproduces the encoder's quantised latents
    y = model.analysis_transform(x.float())
    z = model.hyper_analysis_transform(torch.abs(y))
    z_hat = model.side_entropy_model.quantize(z)
    idx = model.hyper_synthesis_transform(z_hat)[:, :32, :48]
    y_hat = model.entropy_model.decompress(string, idx)
    assert torch.equal(y_hat, torch.round(y))
    bpp = 8 * sum(len(bytes(s)) for s in list(string) + list(side_string)) / (2 * 512 * 768)
    assert 0 < bpp < 24


def test_bmshj2018_compress_many_equals_batch_by_batch():
    """Several batches behind one coder launch per stage (the pipelined lane kernels in index mode for the main
    latent, channel mode for the side latent): same strings and same reconstructions as compress() / decompress()
    one batch at a time (bmshj2018.py:219-264)."""
    torch.manual_seed(4)
    model = tfc.models.BMSHJ2018Model(num_filters=64).cuda().init_compression()
    xs = [torch.from_numpy(synthetic.lowpass_images(3, 200, 136, seed=10 + k)).cuda() for k in range(3)]
    packed = model.compress_many(xs)
    x_hats, oks = model.decompress_many(packed)
    assert all(bool(ok.cpu().all()) for ok in oks)
    for x, p, x_hat in zip(xs, packed, x_hats):
        string, side_string, x_shape, y_shape, z_shape = model.compress(x)
        assert (x_shape, y_shape, z_shape) == tuple(p[2:])
        assert [bytes(s) for s in tfc.fetch_strings(p[0])] == [bytes(s) for s in string]
        assert [bytes(s) for s in tfc.fetch_strings(p[1])] == [bytes(s) for s in side_string]
        assert torch.equal(x_hat, model.decompress(string, side_string, x_shape, y_shape, z_shape))


def test_training_forward_runs():
    torch.manual_seed(3)
    model = tfc.models.BLS2017Model(num_filters=32).cuda()
    x = torch.from_numpy(synthetic.lowpass_images(2, 64, 64)).cuda().float()
    with torch.no_grad():
        loss, bpp, mse = model(x, training=False)
    assert torch.isfinite(loss) and bpp > 0 and mse >= 0


@pytest.mark.parametrize("which", ["bls2017", "bmshj2018", "ms2020"])
def test_tfci_file_round_trip(which, tmp_path):
    """PNG -> .tfci -> PNG through the file helpers (bls2017.py:273-323, ms2020.py:520-568): the container
    parses back to exactly the tensors compress() produced, in compress()'s order, and decodes to the same
    image."""
    from compression_amd import PackedTensors, models, synthetic
    from compression_amd.models import codec_io
    torch.manual_seed(0)
    model = {"bls2017": lambda: models.BLS2017Model(num_filters=64),
             "bmshj2018": lambda: models.BMSHJ2018Model(num_filters=64),
             "ms2020": lambda: models.MS2020Model(num_filters=64, latent_depth=64, hyperprior_depth=32,
                                                  num_slices=2, max_support_slices=1)}[which]()
    model = model.cuda().init_compression()
    img = torch.from_numpy(synthetic.lowpass_images(1, 96, 80, seed=5)[0])
    models.write_png(tmp_path / "in.png", img)
    assert torch.equal(models.read_png(tmp_path / "in.png"), img)
    data = models.compress_file(model, tmp_path / "in.png", tmp_path / "out.tfci")
    direct = model.compress(img.cuda())
    dtypes = codec_io.container_dtypes(model)
    assert len(dtypes) == len(direct)
    if which == "ms2020":
        assert dtypes == [np.int32] * 3 + [bytes] * 3
    for got, want, dtype in zip(PackedTensors(data).unpack(dtypes), direct, dtypes):
        if dtype is bytes:
            assert [bytes(b) for b in got] == [bytes(b) for b in np.asarray(want, dtype=object).reshape(-1)]
        else:
            assert tuple(got.tolist()) == tuple(want)
    x_hat = models.decompress_file(model, tmp_path / "out.tfci", tmp_path / "rec.png")
    assert torch.equal(x_hat.cpu(), model.decompress(*direct)[0].cpu())
    assert torch.equal(models.read_png(tmp_path / "rec.png"), x_hat.cpu())
    assert x_hat.shape == img.shape


def test_bls2017_training_steps_reduce_the_loss():
    """bls2017.py train: forward (analysis -> fused noisy bottleneck -> synthesis), backward through
    every HIP kernel (conv dgrad/wgrad, GDN backward, factorized bits backward) and a few Adam steps."""
    from compression_amd import models
    torch.manual_seed(0)
    model = models.BLS2017Model(lmbda=0.01, num_filters=64).cuda()
    x = torch.from_numpy(synthetic.lowpass_images(4, 64, 64, seed=3)).cuda()
    model(x)                                              # builds the lazily-created kernels
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    losses = []
    for _ in range(12):
        opt.zero_grad()
        loss, bpp, mse = model(x, training=True)
        loss.backward()
        assert all(p.grad is None or torch.isfinite(p.grad).all() for p in model.parameters())
        opt.step()
        losses.append(float(loss.detach()))
    missing = [n for n, p in model.named_parameters() if p.grad is None]
    assert not missing, f"parameters without gradient: {missing}"
    assert np.isfinite(losses).all() and np.mean(losses[-3:]) < np.mean(losses[:3])


def test_ms2020_roundtrip_and_slice_order():
    """ms2020.py:334-420: z string + one string per channel slice; decompress reproduces exactly what the
    encoder's own reconstruction path gives (synthesis of the LRP-corrected slices), and decoding is
    insensitive to nothing but the strings (a wrong slice string changes the output)."""
    torch.manual_seed(4)
    model = tfc.models.MS2020Model(num_filters=64, latent_depth=64, hyperprior_depth=32, num_slices=4,
                                   max_support_slices=2).cuda().init_compression()
    x = torch.from_numpy(synthetic.lowpass_images(2, 96, 80, seed=5)).cuda()       # 80 is not a multiple of 64
    out = model.compress(x)
    x_shape, y_shape, z_shape = out[:3]
    assert x_shape == (96, 80) and y_shape == (6, 5) and z_shape == (2, 2)
    assert len(out) == 4 + 4 and all(s.shape == (2,) for s in out[3:])
    x_hat = model.decompress(*out)
    assert x_hat.shape == (2, 96, 80, 3) and x_hat.dtype == torch.uint8
    # the encoder's view of the reconstruction: quantised slices with the same parameters
    with torch.no_grad():
        y = model.analysis_transform(x.float())
        z = model.hyper_analysis_transform(y)
        z_hat = model.em_z.quantize(z)
        ls, lm = model._hyper_features(z_hat, y_shape)
        slices = []
        for k, ys in enumerate(torch.chunk(y, 4, dim=-1)):
            ms, mu, sigma = model._slice_params(k, lm, ls, slices, y_shape)
            slices.append(model._lrp(k, ms, model.em_y.quantize(ys, loc=mu)))
        ref = model.synthesis_transform(torch.cat(slices, dim=-1))[:, :96, :80, :]
        ref = torch.clamp(torch.round(ref), 0, 255).to(torch.uint8)
    assert torch.equal(x_hat, ref)
    # a second call gives the same strings; swapping two slice strings does not decode to the same image
    again = model.compress(x)
    assert all(bytes(a) == bytes(b) for sa, sb in zip(out[3:], again[3:]) for a, b in zip(sa, sb))
    swapped = list(out)
    swapped[4], swapped[5] = swapped[5], swapped[4]
    try:
        other = model.decompress(*swapped)
        assert not torch.equal(other, x_hat)
    except RuntimeError:
        pass                                          # the decoder's sanity check may already reject it


def test_ms2020_strings_equal_the_oracles():
    """ms2020.py:334-382 pinned to something other than itself: every slice's string is, byte for byte, what the
    reference's index-mode coder (EntropyEncodeIndex + Finalize, range_coder_kernels.cc:217-242, 274-322 — the oracle,
    pinned to the compiled reference in tests/test_oracle.py) writes for the symbols and scale-table indexes the model
    hands to it: round(y_k - mu_k) - cdf_offset[index] with index from sigma_k (continuous_indexed.py:272-289, 355-386),
    the parameters of slice k computed from the decoded slices before it.  The side string likewise, in channel mode
    (continuous_batched.py:370-383)."""
    from oracle import oracle
    port = oracle.best()
    torch.manual_seed(8)
    model = tfc.models.MS2020Model(num_filters=64, latent_depth=64, hyperprior_depth=32, num_slices=4,
                                   max_support_slices=2).cuda().init_compression()
    x = torch.from_numpy(synthetic.lowpass_images(3, 128, 64, seed=9)).cuda()
    out = model.compress(x)
    y_shape, z_shape = out[1], out[2]
    with torch.no_grad():
        y = model.analysis_transform(x.float())
        z = model.hyper_analysis_transform(y)
        em_z, em_y = model.em_z, model.em_y
        # side information: channel mode, symbols = round(z - offset) - cdf_offset[channel]
        zq = em_z.quantize(z)
        qoff = em_z.quantization_offset
        zsym = torch.round(z - qoff if qoff is not None else z).to(torch.int32) - em_z.cdf_offset.cuda()
        want_z, _, _ = port.encode(em_z.cdf.cpu().numpy(), zsym.reshape(3, -1).cpu().numpy())
        assert [bytes(s) for s in out[3]] == want_z
        z_hat = em_z.decompress(out[3], z_shape)
        assert torch.equal(z_hat, zq)
        ls, lm = model._hyper_features(z_hat, y_shape)
        lookup = em_y.cdf.cpu().numpy()
        offsets = em_y.cdf_offset.cuda()
        slices, escapes = [], 0
        for k, ys in enumerate(torch.chunk(y, 4, dim=-1)):
            ms, mu, sigma = model._slice_params(k, lm, ls, slices, y_shape)
            flat = em_y._table_indexes(sigma.contiguous())
            sym = torch.round(ys - mu).to(torch.int32) - offsets[flat.long()]
            sym_h, idx_h = sym.reshape(3, -1).cpu().numpy(), flat.reshape(3, -1).cpu().numpy()
            want, _, _ = port.encode(lookup, sym_h, index=idx_h)
            assert [bytes(s) for s in out[4 + k]] == want, f"slice {k}"
            dec, ok = port.decode(lookup, want, sym_h.shape[1], index=idx_h)
            assert ok.all() and (dec == sym_h).all()
            rows = synthetic.lookup_rows(lookup)
            width = np.array([len(c) - 2 for _, c in rows])
            escapes += int(((sym_h < 0) | (sym_h >= width[idx_h])).sum())
            slices.append(model._lrp(k, ms, em_y.quantize(ys, loc=mu)))
    assert escapes >= 0


def test_ms2020_training_forward_and_backward():
    torch.manual_seed(6)
    # slice depth 32 like the full model (latent_depth 320 / 10 slices); the support tensors are 320 + 32 k
    # channels wide, so the weight gradients of the slice transforms go through the > 256-channel blocking
    model = tfc.models.MS2020Model(num_filters=32, latent_depth=64, hyperprior_depth=32, num_slices=2,
                                   max_support_slices=1).cuda()
    x = torch.from_numpy(synthetic.lowpass_images(2, 64, 64, seed=7)).cuda()
    loss, bpp, mse = model(x, training=True)
    assert torch.isfinite(loss) and bpp > 0 and mse > 0
    loss.backward()
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    assert len(grads) > 20 and all(torch.isfinite(g).all() for g in grads)
    loss_eval, _, _ = model(x, training=False)
    assert torch.isfinite(loss_eval)
