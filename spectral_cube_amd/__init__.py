"""MI355X-native per-spaxel reduction engine for spectral cubes.

Host side: a Python mirror of the reference's SpectralCube operator interface
for the dense hot path (moments, spectral / spatial smoothing, spectral
interpolation, reprojection).  Device side: hand-written HIP kernels for
gfx950 behind the C ABI in include/spcube_hip.h (libspcube_hip.so), bound with
ctypes.  No torch, no CPU fallback.
"""
__version__ = "0.1.0"

from ._lib import HipLibraryError, HipInvalidArgument, HipUnsupported  # noqa: F401
from .cube import (SpectralCube, Projection, VarianceWarning, SmoothingWarning,  # noqa: F401
                   UnitsError, BeamUnitsError, WCSCelestialError, VaryingResolutionSpectralCube, BeamWarning,
                   NonFiniteBeamsWarning, PrecisionWarning)
from .beam import Beam, BeamError  # noqa: F401
from .masks import (BooleanArrayMask, LazyMask, LazyComparisonMask, CompositeMask,  # noqa: F401
                    FunctionMask, InvertedMask)
from .kernels import (Gaussian1DKernel, Gaussian2DKernel, Box1DKernel, Tophat2DKernel,  # noqa: F401
                      CustomKernel)
from .wcs import SimpleWCS  # noqa: F401
