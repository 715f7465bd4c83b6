"""compression_amd — MI355X-native hot path of tensorflow/compression.

Flat namespace in the manner of `tensorflow_compression/__init__.py:17-42`.
The HIP library (libtfc_hip.so) is loaded lazily on first op call; there is no
CPU fallback.
"""
from . import _lib
from .ops import gen_ops
from .ops.gen_ops import *  # noqa: F401,F403
from .ops.math_ops import lower_bound, perturb_and_apply, upper_bound  # noqa: F401
from .ops.padding_ops import same_padding_for_kernel  # noqa: F401
from .ops.round_ops import round_st, soft_round, soft_round_conditional_mean, soft_round_inverse  # noqa: F401
from .distributions import *  # noqa: F401,F403
from .entropy_models import *  # noqa: F401,F403
from .layers import *  # noqa: F401,F403
from .util import PackedTensors  # noqa: F401

__version__ = "0.1.0"
from . import models  # noqa: F401,E402
