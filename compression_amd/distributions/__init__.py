"""Probability models feeding the range-coding tables (the reference's
`python/distributions`)."""
from .base import Distribution, Laplace, Logistic, Normal
from .deep_factorized import DeepFactorized, NoisyDeepFactorized
from .round_adapters import (MonotonicAdapter, NoisyRoundAdapter, NoisyRoundedDeepFactorized, NoisyRoundedNormal,
                             NoisySoftRoundAdapter, NoisySoftRoundedDeepFactorized, NoisySoftRoundedNormal,
                             RoundAdapter, SoftRoundAdapter)
from .helpers import estimate_tails, lower_tail, quantization_offset, upper_tail
from .uniform_noise import (MixtureSameFamily, NoisyLaplace, NoisyLogistic, NoisyLogisticMixture,
                            NoisyMixtureSameFamily, NoisyNormal, NoisyNormalMixture, UniformNoiseAdapter)
