"""Callers of the hot path: the two target model definitions (architecture and shapes
of models/bls2017.py and models/bmshj2018.py; no training loop / dataset plumbing)."""
from . import bls2017, bmshj2018
from .bls2017 import BLS2017Model
from .bmshj2018 import BMSHJ2018Model
from .codec_io import compress_file, decompress_file, read_png, write_png  # noqa: F401
