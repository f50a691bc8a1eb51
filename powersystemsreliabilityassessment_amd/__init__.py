"""MI355X-native non-sequential Monte Carlo HL2 reliability engine (hot path of
Matrixeigs/PowerSystemsReliabilityAssessment: mc_sampling + mc_simulation + the nsqMain loop).

The compute path is the HIP library ``csrc/librelmc.so`` (C ABI in include/relmc.h); this package
only mirrors the reference's operator interface on top of it.  There is no CPU fallback.
"""
from . import case24  # noqa: F401
from ._abi import RELMC_PHYSICAL as PHYSICAL, RELMC_REFERENCE_EMULATE as REFERENCE_EMULATE  # noqa: F401

__all__ = ["case24", "api", "dist", "PHYSICAL", "REFERENCE_EMULATE"]
