"""Scratch: where the wall clock of the all-valid C4 pipeline spatial_smooth -> moment0 goes (cProfile, host side)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from spectral_cube_amd import SpectralCube, Gaussian2DKernel
from spectral_cube_amd.device import DeviceArray, synchronize
from bench_configs_helpers import replicate_planes
nz = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
shape = (nz, 2048, 2048)
rng = np.random.default_rng(2003)
tile = rng.standard_normal((2,) + shape[1:], dtype=np.float32) + 2.0
cube = DeviceArray(shape, np.float32); replicate_planes(cube, tile)
hdr = {"NAXIS": 3, "NAXIS1": 2048, "NAXIS2": 2048, "NAXIS3": nz, "CTYPE1": "RA---TAN", "CTYPE2": "DEC--TAN", "CTYPE3": "VRAD",
       "CRVAL3": 0.0, "CDELT3": 500.0, "CRPIX3": 1.0, "CUNIT3": "m/s", "CDELT1": -1e-4, "CDELT2": 1e-4, "CRPIX1": 1.0,
       "CRPIX2": 1.0, "CRVAL1": 10.0, "CRVAL2": 20.0, "BUNIT": "K"}
sc = SpectralCube.from_device(cube, header=hdr)
k = Gaussian2DKernel(8 / 2.3548200450309493)
for _ in range(3): sc.spatial_smooth(k).moment0()
synchronize(); t0 = time.perf_counter()
for _ in range(5): sc.spatial_smooth(k).moment0()
synchronize(); print("wall per call %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(5): sc.spatial_smooth(k).moment0()
synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
