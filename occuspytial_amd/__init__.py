"""MI355X-native Gibbs sampler for the ICAR spatial occupancy model.

Drop-in for the ``LogitICARGibbs`` path of zoj613/OccuSpytial (v0.2.0): same sampler / chain /
posterior API, the per-iteration work done by hand-written HIP kernels for gfx950 behind a C ABI
(``include/occ_gibbs.h``).  See DESIGN.md.
"""
from .data import Data
from .gibbs import LogitICARGibbs, LogitRSRGibbs

__version__ = '0.2.0'

__all__ = ('LogitICARGibbs', 'LogitRSRGibbs', 'Data', '__version__')
