"""Convolution kernels with the same discretisation as astropy.convolution
(third-party; call sites spectral_cube/dask_spectral_cube.py:880-993).

Only the ``.array`` attribute matters to spectral_smooth / spatial_smooth, and
any object exposing ``.array`` (e.g. a real astropy kernel) is accepted by the
cube methods.  Arrays are compared with astropy's in tests/golden/kernels.npz.
"""
import math

import numpy as np


def _round_up_to_odd_integer(value):
    i = int(math.ceil(value))
    return i + 1 if i % 2 == 0 else i


class Kernel:
    def __init__(self, array):
        self._array = np.asarray(array, dtype=np.float64)

    @property
    def array(self):
        return self._array

    @property
    def shape(self):
        return self._array.shape

    @property
    def dimension(self):
        return self._array.ndim

    def normalize(self):
        self._array = self._array / self._array.sum()
        return self


class Kernel1D(Kernel):
    pass


class Kernel2D(Kernel):
    pass


class CustomKernel(Kernel):
    def __init__(self, array):
        array = np.asarray(array, dtype=np.float64)
        if any(s % 2 == 0 for s in array.shape):
            raise ValueError("Kernel size must be odd in all axes.")
        super().__init__(array)


def _grid(size):
    return np.arange(size, dtype=np.float64) - (size - 1) / 2.0


class Gaussian1DKernel(Kernel1D):
    """astropy.convolution.Gaussian1DKernel (mode='center')."""

    def __init__(self, stddev, x_size=None):
        if hasattr(stddev, "unit"):
            raise TypeError("The convolution kernel should be defined without a unit.")
        stddev = float(stddev)
        size = _round_up_to_odd_integer(8 * stddev) if x_size is None else int(x_size)
        if size % 2 == 0:
            raise ValueError("Kernel size must be odd in all axes.")
        x = _grid(size)
        super().__init__(np.exp(-0.5 * (x / stddev) ** 2) / (math.sqrt(2 * math.pi) * stddev))
        self.stddev = stddev


class Gaussian2DKernel(Kernel2D):
    """astropy.convolution.Gaussian2DKernel (mode='center'); separable when
    theta == 0."""

    def __init__(self, x_stddev, y_stddev=None, theta=0.0, x_size=None, y_size=None):
        x_stddev = float(x_stddev)
        y_stddev = x_stddev if y_stddev is None else float(y_stddev)
        default = _round_up_to_odd_integer(8 * max(x_stddev, y_stddev))
        nx = default if x_size is None else int(x_size)
        ny = (nx if x_size is not None else default) if y_size is None else int(y_size)
        if nx % 2 == 0 or ny % 2 == 0:
            raise ValueError("Kernel size must be odd in all axes.")
        x, y = np.meshgrid(_grid(nx), _grid(ny))
        c, s = math.cos(theta), math.sin(theta)
        a = c * c / (2 * x_stddev ** 2) + s * s / (2 * y_stddev ** 2)
        b = math.sin(2 * theta) / (2 * x_stddev ** 2) - math.sin(2 * theta) / (2 * y_stddev ** 2)
        cc = s * s / (2 * x_stddev ** 2) + c * c / (2 * y_stddev ** 2)
        amp = 1.0 / (2 * math.pi * x_stddev * y_stddev)
        super().__init__(amp * np.exp(-(a * x * x + b * x * y + cc * y * y)))


class Box1DKernel(Kernel1D):
    """astropy.convolution.Box1DKernel (default mode='linear_interp')."""

    def __init__(self, width):
        width = float(width)
        size = _round_up_to_odd_integer(width)
        x = _grid(size)

        def box(t):
            return np.where(np.abs(t) <= width / 2.0, 1.0 / width, 0.0)
        super().__init__(0.5 * (box(x - 0.5) + box(x + 0.5)))


class Tophat2DKernel(Kernel2D):
    """astropy.convolution.Tophat2DKernel (mode='center')."""

    def __init__(self, radius):
        radius = float(radius)
        size = _round_up_to_odd_integer(2 * radius)
        x, y = np.meshgrid(_grid(size), _grid(size))
        super().__init__(np.where(x * x + y * y <= radius * radius, 1.0 / (math.pi * radius ** 2), 0.0))


def kernel_array(kernel, ndim):
    """Extract the tap array of an astropy / own kernel object or ndarray and
    mirror the reference's unit check (dask_spectral_cube.py:908-910)."""
    arr = getattr(kernel, "array", kernel)
    if hasattr(arr, "unit"):
        from .cube import UnitsError
        raise UnitsError("The convolution kernel should be defined without a unit.")
    arr = np.asarray(arr, dtype=np.float64)
    if arr.ndim != ndim:
        raise ValueError("expected a %d-D kernel, got shape %s" % (ndim, arr.shape))
    return arr
