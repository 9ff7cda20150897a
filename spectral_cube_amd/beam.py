"""Minimal Gaussian beam algebra for ``convolve_to``.

The reference delegates to the third-party ``radio_beam`` package (``beam.deconvolve(self.beam)
.as_kernel(pixscale)``, spectral_cube/dask_spectral_cube.py:1445-1447), which is neither vendored nor
installed in this image: the formulas below are restated from its published source
(``radio_beam.utils.deconvolve`` - the classical Gaussian deconvolution of Wild 1970 -,
``Beam.as_kernel`` / ``EllipticalGaussian2DKernel``: support 8 x the larger stddev rounded up to
odd, ``Gaussian2D`` with ``theta = pa + 90 deg``, evaluated at the pixel centres).  PARITY
UNPINNED against radio_beam itself; what is tested is the algebra (covariances add under
convolution) and the image-plane result (tests/test_host_logic.py, tests/test_gpu_cube.py).
"""
import math

import numpy as np

FWHM_TO_SIGMA = 1.0 / math.sqrt(8.0 * math.log(2.0))


class BeamError(Exception):
    """radio_beam.utils.BeamError: the target beam cannot be reached by convolution."""


class Beam:
    """Elliptical Gaussian beam: FWHM major / minor axes and position angle, all in degrees;
    the position angle is measured from +y (north) towards -x (east for the usual RA orientation)."""

    def __init__(self, major, minor=None, pa=0.0):
        self.major = float(major)
        self.minor = float(major if minor is None else minor)
        self.pa = float(pa)
        if self.minor > self.major:
            raise ValueError("Minor axis greater than major axis.")

    @classmethod
    def from_header(cls, header):
        if "BMAJ" not in header:
            return None
        return cls(header["BMAJ"], header.get("BMIN", header["BMAJ"]), header.get("BPA", 0.0))

    @property
    def sr(self):
        """beam area in steradian: pi / (4 ln 2) * major * minor"""
        return math.pi / (4.0 * math.log(2.0)) * math.radians(self.major) * math.radians(self.minor)

    @property
    def isfinite(self):
        return math.isfinite(self.major) and math.isfinite(self.minor) and math.isfinite(self.pa)

    def __eq__(self, other):
        if not isinstance(other, Beam):
            return NotImplemented
        same_pa = abs(((self.pa - other.pa + 90.0) % 180.0) - 90.0) < 1e-9 or abs(self.major - self.minor) < 1e-12 * self.major
        return abs(self.major - other.major) <= 1e-12 * self.major and abs(self.minor - other.minor) <= 1e-12 * self.major and same_pa

    __hash__ = None

    def __repr__(self):
        return "Beam(major=%g deg, minor=%g deg, pa=%g deg)" % (self.major, self.minor, self.pa)

    def covariance(self):
        """2 x 2 covariance (deg^2) in (x, y) pixel-aligned sky axes of the Gaussian this beam is"""
        sa, sb = self.major * FWHM_TO_SIGMA, self.minor * FWHM_TO_SIGMA
        t = math.radians(self.pa) + math.pi / 2.0              # major axis direction, CCW from +x
        c, s = math.cos(t), math.sin(t)
        rot = np.array([[c, -s], [s, c]])
        return rot @ np.diag([sa * sa, sb * sb]) @ rot.T

    def deconvolve(self, other, failure_returns_pointlike=False):
        """the beam that, convolved with *other*, gives this beam (radio_beam.utils.deconvolve)."""
        maj1, min1, pa1 = math.radians(self.major), math.radians(self.minor), math.radians(self.pa)
        maj2, min2, pa2 = math.radians(other.major), math.radians(other.minor), math.radians(other.pa)
        alpha = ((maj1 * math.cos(pa1)) ** 2 + (min1 * math.sin(pa1)) ** 2 -
                 (maj2 * math.cos(pa2)) ** 2 - (min2 * math.sin(pa2)) ** 2)
        beta = ((maj1 * math.sin(pa1)) ** 2 + (min1 * math.cos(pa1)) ** 2 -
                (maj2 * math.sin(pa2)) ** 2 - (min2 * math.cos(pa2)) ** 2)
        gamma = 2.0 * ((min1 ** 2 - maj1 ** 2) * math.sin(pa1) * math.cos(pa1) -
                       (min2 ** 2 - maj2 ** 2) * math.sin(pa2) * math.cos(pa2))
        s = alpha + beta
        t = math.sqrt((alpha - beta) ** 2 + gamma ** 2)
        limit = 0.1 * min(maj1, min1, maj2, min2) ** 2 * 1e-6   # (radio_beam: numerical slack around zero)
        if alpha < -limit or beta < -limit or s < t - limit:
            if failure_returns_pointlike:
                return Beam(0.0, 0.0, 0.0)
            raise BeamError("Beam could not be deconvolved")
        new_major = math.sqrt(0.5 * (s + t))
        new_minor = math.sqrt(max(0.5 * (s - t), 0.0))
        if abs(gamma) + abs(alpha - beta) == 0.0:
            new_pa = 0.0
        else:
            new_pa = 0.5 * math.atan2(-gamma, alpha - beta)
        return Beam(math.degrees(new_major), math.degrees(new_minor), math.degrees(new_pa))

    def as_kernel(self, pixscale_deg, support_scaling=8.0):
        """sampled elliptical Gaussian on the pixel grid (EllipticalGaussian2DKernel.array)."""
        smaj = self.major * FWHM_TO_SIGMA / pixscale_deg
        smin = self.minor * FWHM_TO_SIGMA / pixscale_deg
        if not (smaj > 0.0 and smin > 0.0):
            raise BeamError("cannot build a kernel for a point-like beam")
        size = int(math.ceil(support_scaling * max(smaj, smin)))
        size += 1 - size % 2                                   # round up to odd
        h = size // 2
        yy, xx = np.mgrid[-h:h + 1, -h:h + 1].astype(np.float64)
        theta = math.radians(self.pa) + math.pi / 2.0
        cost2, sint2, sin2t = math.cos(theta) ** 2, math.sin(theta) ** 2, math.sin(2.0 * theta)
        xs2, ys2 = smaj * smaj, smin * smin
        a = 0.5 * (cost2 / xs2 + sint2 / ys2)
        b = 0.5 * (sin2t / xs2 - sin2t / ys2)
        c = 0.5 * (sint2 / xs2 + cost2 / ys2)
        amp = 1.0 / (2.0 * math.pi * smaj * smin)
        return amp * np.exp(-(a * xx * xx + b * xx * yy + c * yy * yy))
