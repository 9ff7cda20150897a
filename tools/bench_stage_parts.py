"""Scratch: where a dask chunk's staging time goes (host copy of a (nz, cy, cx) window into pinned memory, H2D, kernel, D2H)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spectral_cube_amd import dask_adapter as A
from spectral_cube_amd.device import synchronize
n = 1024
d = np.tile(np.random.default_rng(0).standard_normal((n, 16, n)).astype(np.float32), (1, n // 16, 1))
chunk = d[:, 256:512, 512:768]
cen = np.arange(n, dtype=np.float64) * 500.0
def t(fn, reps=5):
    fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3
buf = np.empty(chunk.shape, np.float32)
print("np.copyto window -> pageable contiguous      %.1f ms" % t(lambda: np.copyto(buf, chunk)))
def stage():
    dev, st = A._stage(chunk, 0); st.stream.synchronize()
print("_stage (copy pieces -> pinned + H2D), waited  %.1f ms" % t(stage))
os.environ["X"] = "1"
st = A._ctx(0)
ptr = st.pinned(chunk.size * 4)
import ctypes as C
view = np.frombuffer((C.c_byte * (chunk.size * 4)).from_address(ptr), dtype=np.float32, count=chunk.size).reshape(chunk.shape)
print("np.copyto window -> pinned, one thread        %.1f ms" % t(lambda: np.copyto(view, chunk)))
print("np.copyto contiguous -> pinned, one thread    %.1f ms" % t(lambda: np.copyto(view, buf)))
f = A.Moments012Chunk(cen, 500.0)
print("Moments012Chunk(chunk) end to end             %.1f ms" % t(lambda: f(chunk)))
cont = np.ascontiguousarray(chunk)
print("Moments012Chunk(contiguous copy) end to end   %.1f ms" % t(lambda: f(cont)))
