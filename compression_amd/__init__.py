"""compression_amd — MI355X-native hot path of tensorflow/compression.

Flat namespace in the manner of `tensorflow_compression/__init__.py:17-42`.
The HIP library (libtfc_hip.so) is loaded lazily on first op call; there is no
CPU fallback.
"""
from . import _lib
from .ops import gen_ops
from .ops.gen_ops import *  # noqa: F401,F403

__version__ = "0.1.0"
