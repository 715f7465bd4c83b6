from .functional import conv2d_down, conv2d_up, gdn_forward  # noqa: F401
