"""Scratch: C5 reproject (4096 x 1024^2, 30-degree rotated grid) only, for kernel experiments."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from spectral_cube_amd import ops, synth
from spectral_cube_amd.device import DeviceArray, Event, synchronize
from spectral_cube_amd.wcs import SimpleWCS, reproject_pixel_map
from test_gpu_fullsize import _replicate_rows

nz = int(os.environ.get("NZ", 4096))
shape = (nz, 1024, 1024)
tile = synth.gaussian_line_cube((shape[0], 2, shape[2]), 2004, chunk_rows=2)
cube = DeviceArray(shape, np.float32); _replicate_rows(cube, tile, 4)
hdr = {"CTYPE1": "RA---TAN", "CTYPE2": "DEC--TAN", "CRVAL1": 150.0, "CRVAL2": 2.0, "CRPIX1": 512.5, "CRPIX2": 512.5,
       "CDELT1": -1 / 3600, "CDELT2": 1 / 3600, "NAXIS": 2}
out = DeviceArray(shape, np.float32)
ref = None
for deg in (30.0,):
    c, s_ = np.cos(np.radians(deg)), np.sin(np.radians(deg))
    w_in, w_out = SimpleWCS(hdr, naxis=2), SimpleWCS(dict(hdr, PC1_1=c, PC1_2=-s_, PC2_1=s_, PC2_2=c), naxis=2)
    xs, ys = reproject_pixel_map(w_in, w_out, (1024, 1024))
    for label, env in [("tile 64", {"SPC_BILINEAR_TILE": "64"})] + [("tile 64 z%d" % z, {"SPC_BILINEAR_TILE": "64", "SPC_BILINEAR_ZCHUNK": str(z)}) for z in (8, 16, 32, 64, 128)] + \
                      [("tile 32 z%d" % z, {"SPC_BILINEAR_TILE": "32", "SPC_BILINEAR_ZCHUNK": str(z)}) for z in (16, 64, 256)]:
        for k in ("SPC_BILINEAR_ZCHUNK", "SPC_BILINEAR_TILE"): os.environ.pop(k, None)
        os.environ.update(env)
        ts = []
        for i in range(5):
            e0, e1 = Event(), Event()
            e0.record(); ops.resample_bilinear(cube, xs, ys, out=out); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_ms(e1))
        vox = np.prod(shape, dtype=np.int64)
        print("%4.0f deg %-14s median %.3f ms  %.0f GB/s  %.1f%%" % (deg, label, np.median(ts[1:]), vox * 8 / np.median(ts[1:]) / 1e6, vox * 8 / np.median(ts[1:]) / 1e6 / 80), flush=True)
