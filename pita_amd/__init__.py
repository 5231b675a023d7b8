"""pita_amd -- MI355X-native inference-time annealed-SDE Boltzmann sampler (drop-in for the
sampling path of taraak/pita).  Host classes mirror the reference plug-in API; all arithmetic
on walkers runs in hand-written HIP kernels behind the C ABI of libpita_hip.so."""
from . import _lib  # noqa: F401
from .annealing_factor_schedules import (ConstantAnnealingFactorSchedule, LinearAnnealingFactorSchedule,  # noqa: F401
                                         SigmoidAnnealingFactorSchedule)
from .base_prior import Prior  # noqa: F401
from .dw4_energy import MultiDoubleWellEnergy  # noqa: F401
from .egnn_temp_conditioned import EGNN_dynamics  # noqa: F401
from .gmm_energy import GMM  # noqa: F401
from .lennardjones_energy import LennardJonesEnergy  # noqa: F401
from .noise_schedules import ElucidatingNoiseSchedule, GeometricNoiseSchedule  # noqa: F401
from .score_net import ScoreNet  # noqa: F401
from .sde_integration import WeightedSDEIntegrator  # noqa: F401
from .sdes import SDETerms, VEReverseSDE  # noqa: F401

__version__ = "0.1.0"
