import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, cProfile, pstats
from spectral_cube_amd import SpectralCube
from spectral_cube_amd.device import DeviceArray, synchronize
shape = (1024, 1024, 1024)
dev = DeviceArray.zeros(shape, np.float32)
hdr = {"CTYPE1": "RA---TAN", "CTYPE2": "DEC--TAN", "CTYPE3": "VRAD", "CDELT1": -1 / 3600, "CDELT2": 1 / 3600, "CDELT3": 0.5,
       "CUNIT3": "km/s", "CRPIX1": 512.5, "CRPIX2": 512.5, "CRPIX3": 1, "CRVAL1": 150.0, "CRVAL2": 2.0, "CRVAL3": -256.0, "BUNIT": "K"}
cube = SpectralCube.from_device(dev, header=hdr)
v = cube.spectral_axis
f = lambda: cube.spectral_interpolate(np.linspace(v[0], v[-1], 2048))._device_data()
r = f(); synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(2): r = f(); synchronize()
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
