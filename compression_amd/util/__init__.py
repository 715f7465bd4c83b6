from .packed_tensors import PackedTensors  # noqa: F401
