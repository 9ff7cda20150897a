"""Stub of the third-party `radio_beam` package (absent in this image).

TEST INFRASTRUCTURE ONLY: lets the *reference* package import inside the build
container so golden vectors can be generated (oracle/ref_env/bootstrap.py).
Beams are not on the hot path; every entry point that
would need real beam maths raises.
"""
from . import utils, beam  # noqa: F401


class Beam:
    def __init__(self, *a, **k):
        raise NotImplementedError("radio_beam stub")

    @classmethod
    def from_fits_header(cls, hdr, *a, **k):
        from .utils import NoBeamException
        raise NoBeamException("radio_beam stub: no beam support")


class Beams(list):
    pass


class EllipticalGaussian2DKernel:  # pragma: no cover
    pass
