from .functional import conv2d_down, conv2d_up, gdn_backward, gdn_forward  # noqa: F401
from .gdn import GDN  # noqa: F401
from .signal_conv import SignalConv2D  # noqa: F401
from .initializers import IdentityInitializer  # noqa: F401
from .parameters import GDNParameter, Parameter, RDFTParameter  # noqa: F401
from .soft_round import SoftRound, SoftRoundConditionalMean  # noqa: F401

__all__ = ["conv2d_down", "conv2d_up", "gdn_backward", "gdn_forward", "GDN", "SignalConv2D", "SoftRound",
           "SoftRoundConditionalMean", "IdentityInitializer", "Parameter", "RDFTParameter", "GDNParameter"]
