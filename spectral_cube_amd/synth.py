"""Seeded synthetic position-position-velocity cubes.

Pure numpy, deterministic for a given seed; used by the tests, the golden
vector generator and bench.py.  Nothing here is shipped from the reference.
"""
import hashlib

import numpy as np

SEEDS = {"C1": 1234, "C2": 2001, "C3": 2002, "C4": 2003, "C5": 2004}


def gaussian_line_cube(shape, seed, noise=0.5, dtype=np.float32, chunk_rows=64):
    """fp32 cube with one Gaussian emission line per spaxel plus noise.

    amplitude U(1,4), centre nz/2 + U(-nz/8, nz/8) channels, sigma U(4,12)
    channels, noise N(0, noise).  Generated strip by strip (rows of y) so that
    multi-GiB cubes never need a float64 temporary of the full size.
    """
    nz, ny, nx = shape
    rng = np.random.default_rng(seed)
    out = np.empty(shape, dtype=dtype)
    z = np.arange(nz, dtype=np.float32)[:, None, None]
    for y0 in range(0, ny, chunk_rows):
        y1 = min(ny, y0 + chunk_rows)
        n = y1 - y0
        amp = rng.uniform(1.0, 4.0, size=(n, nx)).astype(np.float32)
        cen = (nz / 2.0 + rng.uniform(-nz / 8.0, nz / 8.0, size=(n, nx))).astype(np.float32)
        sig = rng.uniform(4.0, 12.0, size=(n, nx)).astype(np.float32)
        line = amp * np.exp(-0.5 * ((z - cen) / sig) ** 2)
        line += rng.standard_normal(size=(nz, n, nx), dtype=np.float32) * np.float32(noise)
        out[:, y0:y1, :] = line
    return out


def boolean_mask(data, seed, noise=0.5, flip=0.01):
    """uint8 include mask = (data > 2*noise) XOR a seeded 1 % random flip, with
    one fully masked 8x8 spaxel block (exercises the all-bad -> NaN path)."""
    rng = np.random.default_rng(seed + 7919)
    m = data > np.float32(2.0 * noise)
    nz, ny, nx = data.shape
    for z0 in range(0, nz, 64):           # strip-wise to bound temporaries
        z1 = min(nz, z0 + 64)
        m[z0:z1] ^= rng.random(size=(z1 - z0, ny, nx), dtype=np.float32) < flip
    m[:, :min(8, ny), :min(8, nx)] = False
    return m.view(np.uint8)


def add_nan_block(data, y0=8, x0=8, size=8):
    """NaN *input* block (not masked): exercises NaN data under the mask."""
    data[:, y0:y0 + size, x0:x0 + size] = np.nan
    return data


def spectral_axis(nz, dv=500.0):
    """Linear spectral axis in m/s: channel 0 at -dv*nz/2 so the absolute
    moment-1 map crosses zero."""
    return -dv * nz / 2.0 + dv * np.arange(nz, dtype=np.float64)


def sha256(a):
    return hashlib.sha256(np.ascontiguousarray(a).view(np.uint8)).hexdigest()
