"""Scratch: where the wall clock of spatial_smooth -> moment0 (all valid: algebraic path) goes; maps are 2048^2."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from spectral_cube_amd import ops, _lib, SpectralCube, Gaussian2DKernel
from spectral_cube_amd.device import DeviceArray, synchronize
from test_gpu_fullsize import _replicate_planes
shape = tuple(int(s) for s in (sys.argv[1:4] or (512, 2048, 2048)))
rng = np.random.default_rng(2003)
tile = rng.standard_normal((2,) + shape[1:], dtype=np.float32) + 2.0
cube = DeviceArray(shape, np.float32); _replicate_planes(cube, tile, 4)
hdr = {"NAXIS": 3, "NAXIS1": shape[2], "NAXIS2": shape[1], "NAXIS3": shape[0], "CTYPE1": "RA---TAN", "CTYPE2": "DEC--TAN",
       "CTYPE3": "VRAD", "CRVAL3": 0.0, "CDELT3": 500.0, "CRPIX3": 1.0, "CUNIT3": "m/s", "CDELT1": -1e-4, "CDELT2": 1e-4,
       "CRPIX1": 1.0, "CRPIX2": 1.0, "CRVAL1": 10.0, "CRVAL2": 20.0, "BUNIT": "K"}
sc = SpectralCube.from_device(cube, header=hdr)
kobj = Gaussian2DKernel(8 / 2.3548200450309493)
def wall(fn, n=5):
    fn(); fn(); synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("spatial_smooth -> moment0   %.2f ms" % wall(lambda: sc.spatial_smooth(kobj).moment0()))
print("spatial_smooth -> moment1   %.2f ms" % wall(lambda: sc.spatial_smooth(kobj).moment1()))
print("moment0 (no smoothing)      %.2f ms" % wall(lambda: sc.moment0()))
r = sc._moment_device(("s0", "nvalid"))
print("  _moment_device(s0,nvalid) %.2f ms" % wall(lambda: sc._moment_device(("s0", "nvalid"))))
print("  nvalid.get().min()        %.2f ms" % wall(lambda: int(r["nvalid"].get().min())))
print("  map_conv2d (device)       %.2f ms" % wall(lambda: ops.map_conv2d(r["s0"], kobj.array)))
c0d = ops.map_conv2d(r["s0"], kobj.array)
print("  c0.get()                  %.2f ms" % wall(lambda: c0d.get()))
c0 = c0d.get()
print("  isfinite/!=0 checks       %.2f ms" % wall(lambda: np.all(np.isfinite(c0)) and np.all(c0 != 0.0)))
print("  dv * c0                   %.2f ms" % wall(lambda: 2.0 * c0))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3): sc.spatial_smooth(kobj).moment0()
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
