"""Scratch (conda python with dask): what of the map_blocks wall clock is dask's, what the chunk function's."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import dask, dask.array as da
from spectral_cube_amd.dask_adapter import Moments012Chunk
n = 1024
d = np.tile(np.random.default_rng(0).standard_normal((n, 16, n)).astype(np.float32), (1, n // 16, 1))
cen = np.arange(n, dtype=np.float64) * 500.0
cy = cx = 256
arr = da.from_array(d, chunks=(-1, cy, cx))
f = Moments012Chunk(cen, 500.0)
chunk = d[:, 256:512, 512:768]
def t(fn, reps=3):
    fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3
print("numpy", np.__version__, "dask", dask.__version__)
print("f(chunk) standalone                         %.1f ms" % t(lambda: f(chunk)))
print("16 x f(window) in a plain loop              %.1f ms" % t(lambda: [f(d[:, y:y + cy, x:x + cx]) for y in range(0, n, cy) for x in range(0, n, cx)]))
def graph(fn):
    return da.map_blocks(fn, arr, dtype=np.float64, drop_axis=[0], new_axis=[0], chunks=((3,), arr.chunks[1], arr.chunks[2]))
for w in (1, 8):
    with dask.config.set(scheduler="threads", num_workers=w):
        print("workers=%d  trivial chunk function (zeros)     %.1f ms" % (w, t(lambda: graph(lambda c: np.zeros((3,) + c.shape[1:])).compute())))
        print("workers=%d  chunk function touching c.sum()    %.1f ms" % (w, t(lambda: graph(lambda c: np.zeros((3,) + c.shape[1:]) + c[0, 0, 0]).compute())))
        print("workers=%d  Moments012Chunk                    %.1f ms" % (w, t(lambda: graph(f).compute())))
arr2 = da.from_array(d, chunks=(-1, cy, cx), asarray=False) if False else None
