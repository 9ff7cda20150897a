"""Scratch/report: the cube-level dask entry (dask_adapter.DaskCubeOps: windows of the dask array straight into the strip
pipeline's pinned ring) next to the per-chunk seam (map_blocks + chunk functions) - run under the interpreter that has dask:
LD_PRELOAD=/usr/lib/x86_64-linux-gnu/libstdc++.so.6 /opt/conda/bin/python3.9 -B tools/bench_dask_cube.py [n]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "oracle"))
import numpy as np
import dask, dask.array as da
from spectral_cube_amd.dask_adapter import DaskCubeOps, Moments012Chunk, SpectralSmoothChunk
from spectral_cube_amd.kernels import Gaussian1DKernel
from spectral_cube_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
d = np.tile(synth.gaussian_line_cube((n, 16, n), 5, chunk_rows=16), (1, n // 16, 1))
vox = d.size
cy = cx = max(64, n // 4)
arr = da.from_array(d, chunks=(-1, cy, cx))
hdr = {"CTYPE1": "RA---TAN", "CTYPE2": "DEC--TAN", "CTYPE3": "VRAD", "CDELT1": -1e-3, "CDELT2": 1e-3, "CDELT3": 500.0, "CUNIT3": "m/s",
       "CRPIX1": 1, "CRPIX2": 1, "CRPIX3": 1, "CRVAL1": 10.0, "CRVAL2": 20.0, "CRVAL3": 0.0}
k1 = Gaussian1DKernel(4)


def best(fn, reps=3):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); ts.append(time.perf_counter() - t0)
    return min(ts), out


ops_ = DaskCubeOps(arr, hdr)
t, m = best(lambda: ops_.moments012())
print("cube-level moments 0+1+2 of %d^3 (dask array -> strip pipeline)        %8.1f ms  %6.1f GB/s in" % (n, t * 1e3, vox * 4 / t / 1e9), flush=True)
t, sm = best(lambda: ops_.spectral_smooth(k1))
print("cube-level spectral_smooth(33 taps) -> dask array over the host sink  %8.1f ms  %6.1f GB/s each way" % (t * 1e3, vox * 4 / t / 1e9), flush=True)
cen = np.arange(n, dtype=np.float64) * 500.0
with dask.config.set(scheduler="threads", num_workers=8):
    t, m2 = best(lambda: da.map_blocks(Moments012Chunk(cen, 500.0, 0.0), arr, dtype=np.float64, drop_axis=[0], new_axis=[0],
                                       chunks=((3,), arr.chunks[1], arr.chunks[2])).compute())
    print("per-chunk seam  moments 0+1+2 (map_blocks, 8 workers)                 %8.1f ms  %6.1f GB/s in" % (t * 1e3, vox * 4 / t / 1e9), flush=True)
    t, sm2 = best(lambda: da.map_blocks(SpectralSmoothChunk(k1.array), arr, dtype=arr.dtype).compute())
    print("per-chunk seam  spectral_smooth (map_blocks, 8 workers)                %8.1f ms  %6.1f GB/s each way" % (t * 1e3, vox * 4 / t / 1e9), flush=True)
assert np.array_equal(np.asarray(sm), sm2, equal_nan=True), "the two routes disagree"
for a, b in zip(m, m2):
    assert np.allclose(a, b, rtol=0, atol=1e-9 * np.nanmax(np.abs(b)), equal_nan=True)
print("routes agree")
