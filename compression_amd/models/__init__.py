"""Callers of the hot path: the model definitions (architecture and shapes of models/bls2017.py,
models/bmshj2018.py and models/ms2020.py; no training loop / dataset plumbing)."""
from . import bls2017, bmshj2018, ms2020
from .bls2017 import BLS2017Model
from .bmshj2018 import BMSHJ2018Model
from .ms2020 import MS2020Model
from .codec_io import compress_file, decompress_file, read_png, write_png  # noqa: F401
