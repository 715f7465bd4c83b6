"""Operator-level API (the reference's `python/ops`)."""
from . import gen_ops, math_ops, padding_ops, round_ops
from .gen_ops import *  # noqa: F401,F403
