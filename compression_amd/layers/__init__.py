from .functional import gdn_forward  # noqa: F401
