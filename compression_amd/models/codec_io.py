"""File-level compress / decompress of the models and their command line
(models/bls2017.py:273-323, models/bmshj2018.py:348-398, models/ms2020.py:520-568): PNG in, `.tfci` out and
back.  The container is the reference's PackedTensors layout, in the order the model's compress() returns —
bls2017: [string, x_shape, y_shape]; bmshj2018: [string, side_string, x_shape, y_shape, z_shape]; ms2020:
[x_shape, y_shape, z_shape, z_string, y_string_0 ...] — so files are interchangeable with ones a reference
model of the same weights would write."""
from __future__ import annotations

import argparse

import numpy as np
import torch

from ..util import PackedTensors

__all__ = ["read_png", "write_png", "compress_file", "decompress_file", "container_dtypes", "load_checkpoint",
           "main"]


def read_png(filename) -> torch.Tensor:
    """uint8 [H, W, 3] (bls2017.py:39-43)."""
    from PIL import Image
    with Image.open(filename) as im:
        return torch.from_numpy(np.asarray(im.convert("RGB"), dtype=np.uint8).copy())


def write_png(filename, image) -> None:
    """bls2017.py:46-50."""
    from PIL import Image
    arr = image.detach().cpu().numpy() if isinstance(image, torch.Tensor) else np.asarray(image)
    Image.fromarray(arr.astype(np.uint8), "RGB").save(filename, format="PNG")


def _pack(tensors) -> PackedTensors:
    packed = PackedTensors()
    packed.pack([np.asarray(t, dtype=object) if isinstance(t, np.ndarray) and t.dtype == object
                 else np.asarray(t, dtype=np.int32) for t in tensors])
    return packed


def container_dtypes(model):
    """The dtypes of decompress()'s arguments, in order (the reference reads them off
    `model.decompress.input_signature`, bls2017.py:314, ms2020.py:560): a model either names them itself
    (`container_dtypes`, ms2020) or has its strings first and its shapes after them."""
    own = getattr(model, "container_dtypes", None)
    if own is not None:
        return list(own)
    return [bytes] * model.num_strings + [np.int32] * (model.num_packed - model.num_strings)


def compress_file(model, input_file, output_file, verbose=False):
    """bls2017.py:273-307: one image -> .tfci; returns the container bytes."""
    x = read_png(input_file)
    device = next(model.parameters()).device
    tensors = model.compress(x.to(device))
    packed = _pack(tensors)
    data = packed.string
    with open(output_file, "wb") as f:
        f.write(data)
    if verbose:
        x_hat = model.decompress(*tensors)[0].float().cpu()
        mse = torch.mean((x.float() - x_hat) ** 2).item()
        psnr = 10.0 * np.log10(255.0 ** 2 / mse) if mse > 0 else float("inf")
        print(f"Mean squared error: {mse:0.4f}")
        print(f"PSNR (dB): {psnr:0.2f}")
        print(f"Bits per pixel: {len(data) * 8 / (x.shape[0] * x.shape[1]):0.4f}")
    return data


def decompress_file(model, input_file, output_file=None):
    """bls2017.py:310-323: .tfci -> uint8 [H, W, 3] (and a PNG if output_file is given)."""
    with open(input_file, "rb") as f:
        packed = PackedTensors(f.read())
    dtypes = container_dtypes(model)
    tensors = packed.unpack(dtypes)
    tensors = [t if d is bytes else tuple(int(v) for v in t) for t, d in zip(tensors, dtypes)]
    x_hat = model.decompress(*tensors)[0]
    if output_file is not None:
        write_png(output_file, x_hat)
    return x_hat


def load_checkpoint(model, state_dict):
    """Loads a state_dict into `model` and leaves it ready to compress / decompress.

    A checkpoint written AFTER `init_compression()` carries the range-coding tables (`_cdf`,
    `_cdf_offset`, the quantization offset): like the reference, whose saved model holds them, they are
    LOADED, never regenerated on the receiving side (continuous_base.py:175-184) — regenerated tables are
    only bit-identical when the prior evaluates identically on both machines and software stacks.  The
    entropy models are created first (so the buffers exist), their buffers take the stored shapes, then
    everything is loaded strictly.  A checkpoint without tables gets them built from its prior."""
    has_tables = any(k.rsplit(".", 1)[-1] in ("_cdf", "_cdf_offset") for k in state_dict)
    if not has_tables:
        model.load_state_dict(state_dict)
        return model.init_compression()
    model.init_compression()
    own = dict(model.named_buffers())
    for key, value in state_dict.items():
        if key in own and own[key].shape != value.shape:
            mod_name, _, buf_name = key.rpartition(".")
            module = model.get_submodule(mod_name) if mod_name else model
            module.register_buffer(buf_name, torch.zeros_like(value, device=own[key].device))
    model.load_state_dict(state_dict)
    return model


def main(model_cls, argv=None):
    """`python -m compression_amd.models.bls2017 compress in.png out.tfci` / `decompress in.tfci out.png`.
    --model_path takes a torch state_dict (the reference loads a saved Keras model); without
    it the model keeps its initialisers, which is enough to exercise the path."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--model_path", default=None)
    ap.add_argument("--num_filters", type=int, default=192)
    ap.add_argument("--verbose", "-V", action="store_true")
    ap.add_argument("--seed", type=int, default=0, help="initialiser seed when no --model_path is given")
    sub = ap.add_subparsers(dest="command", required=True)
    for name in ("compress", "decompress"):
        sp = sub.add_parser(name)
        sp.add_argument("input_file")
        sp.add_argument("output_file", nargs="?")
    args = ap.parse_args(argv)
    torch.manual_seed(args.seed)
    model = model_cls(num_filters=args.num_filters).cuda()
    if args.model_path:
        model = load_checkpoint(model, torch.load(args.model_path, map_location="cpu"))
    else:
        model = model.init_compression()
    if args.command == "compress":
        compress_file(model, args.input_file, args.output_file or args.input_file + ".tfci", args.verbose)
    else:
        decompress_file(model, args.input_file, args.output_file or args.input_file + ".png")
    return 0


