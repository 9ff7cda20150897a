"""Scratch: wall clock of SpectralCube-level calls on a device-resident 1024^3 cube (kernel + host glue + result
download), to find host-side overheads worth moving to the device."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from spectral_cube_amd import SpectralCube, BooleanArrayMask, Gaussian1DKernel, Gaussian2DKernel, synth
from spectral_cube_amd.device import DeviceArray, synchronize
from test_gpu_fullsize import _replicate_rows
warnings.simplefilter("ignore")
shape = (1024, 1024, 1024)
tile = synth.gaussian_line_cube((shape[0], 8, shape[2]), 2001, chunk_rows=8)
dev = DeviceArray(shape, np.float32); _replicate_rows(dev, tile, 4)
hdr = {"CTYPE1": "RA---TAN", "CTYPE2": "DEC--TAN", "CTYPE3": "VRAD", "CDELT1": -1 / 3600, "CDELT2": 1 / 3600, "CDELT3": 0.5,
       "CUNIT3": "km/s", "CRPIX1": 512.5, "CRPIX2": 512.5, "CRPIX3": 1, "CRVAL1": 150.0, "CRVAL2": 2.0, "CRVAL3": -256.0, "BUNIT": "K"}
cube = SpectralCube.from_device(dev, header=hdr)
cube.allow_huge_operations = True          # (the reference's guard for reproject / convolve_to above 1e8 voxels)


def wall(label, fn, n=4):
    ts = []
    for _ in range(n):
        synchronize(); t0 = time.perf_counter(); r = fn(); synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("%-64s first %8.2f ms   best of rest %8.2f ms" % (label, ts[0], min(ts[1:])), flush=True)


wall("cube.moment0()", lambda: cube.moment0())
wall("cube.moment1()", lambda: cube.moment1())
wall("cube.moments012()", lambda: cube.moments012())
wall("cube.argmax(axis=0)", lambda: cube.argmax(axis=0))
wall("cube.statistics()", lambda: cube.statistics())
wall("cube.mean(axis=(1, 2))", lambda: cube.mean(axis=(1, 2)))
wall("cube.median(axis=0)", lambda: cube.median(axis=0))
wall("cube.median()", lambda: cube.median())
wall("cube.spectral_smooth(G1(4)).moment1()  [fused]", lambda: cube.spectral_smooth(Gaussian1DKernel(4)).moment1())
wall("cube.spectral_smooth(G1(4))  [materialised]", lambda: cube.spectral_smooth(Gaussian1DKernel(4))._device_data())
wall("cube.spatial_smooth(G2(3.4)).moment0()  [algebraic]", lambda: cube.spatial_smooth(Gaussian2DKernel(8 / 2.35482)).moment0())
c, s_ = np.cos(np.radians(30)), np.sin(np.radians(30))
tgt = dict(hdr, PC1_1=c, PC1_2=-s_, PC2_1=s_, PC2_2=c, NAXIS=3, NAXIS1=1024, NAXIS2=1024, NAXIS3=1024)
wall("cube.reproject(rotated 30 deg)", lambda: cube.reproject(tgt)._device_data(), n=3)
v = cube.spectral_axis
wall("cube.spectral_interpolate(2048 channels)", lambda: cube.spectral_interpolate(np.linspace(v[0], v[-1], 2048))._device_data(), n=3)
masked = cube.with_mask(cube > 0.5)
wall("(cube > 0.5 mask).moment0()", lambda: masked.moment0())
if os.environ.get("SPC_WALL_PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    r = cube.spectral_interpolate(np.linspace(v[0], v[-1], 2048))._device_data(); synchronize()
    pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(8)
