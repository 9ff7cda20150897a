"""Scratch: cost of hipMalloc / hipFree through spc_malloc / spc_free at cube sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spectral_cube_amd.device import DeviceArray, synchronize
for gib in (0.25, 1, 4, 16, 32):
    n = int(gib * (1 << 30))
    ta, tf = [], []
    keep = []
    for i in range(6):
        synchronize(); t0 = time.perf_counter()
        a = DeviceArray((n,), np.uint8)
        t1 = time.perf_counter()
        a.free()
        t2 = time.perf_counter()
        ta.append((t1 - t0) * 1e3); tf.append((t2 - t1) * 1e3)
    print("%5.2f GiB  malloc ms %s   free ms %s" % (gib, ["%.1f" % t for t in ta], ["%.1f" % t for t in tf]), flush=True)
# does a first-touch cost exist on top (memset of a fresh allocation vs a reused one)?
from spectral_cube_amd import _lib
import ctypes as C
a = DeviceArray((4 << 30,), np.uint8)
for i in range(3):
    synchronize(); t0 = time.perf_counter()
    _lib.call("spc_memset", 0, C.c_void_p(a.ptr), 0, a.nbytes, None); synchronize()
    print("memset 4 GiB pass %d: %.2f ms" % (i, (time.perf_counter() - t0) * 1e3), flush=True)
