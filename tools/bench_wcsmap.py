"""Scratch: reprojection pixel map, host (numpy) vs device, 1024^2 and 2048^2 target pixels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spectral_cube_amd import ops
from spectral_cube_amd.device import synchronize
from spectral_cube_amd.wcs import SimpleWCS, reproject_pixel_map
hdr = {"CTYPE1": "RA---TAN", "CTYPE2": "DEC--TAN", "CRVAL1": 150.0, "CRVAL2": 2.0, "CRPIX1": 512.5, "CRPIX2": 512.5,
       "CDELT1": -1 / 3600, "CDELT2": 1 / 3600, "NAXIS": 2}
c, s_ = np.cos(np.radians(30)), np.sin(np.radians(30))
w_in, w_out = SimpleWCS(hdr, naxis=2), SimpleWCS(dict(hdr, PC1_1=c, PC1_2=-s_, PC2_1=s_, PC2_2=c), naxis=2)
for n in (1024, 2048):
    t0 = time.perf_counter(); ex, ey = reproject_pixel_map(w_in, w_out, (n, n)); th = time.perf_counter() - t0
    ops.wcs_pixel_map(w_in, w_out, (n, n)); synchronize()
    t0 = time.perf_counter(); dx, dy = ops.wcs_pixel_map(w_in, w_out, (n, n)); synchronize(); td = time.perf_counter() - t0
    err = max(np.abs(dx.get() - ex).max(), np.abs(dy.get() - ey).max())
    print("%d^2 pixels: host numpy %.1f ms, device %.3f ms, max |diff| %.2e px" % (n, th * 1e3, td * 1e3, err), flush=True)
