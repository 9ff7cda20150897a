"""Scratch: host <-> HBM rates of DeviceArray.from_numpy (pageable hipMemcpy) and DeviceArray.get() for cube-sized arrays."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spectral_cube_amd.device import DeviceArray, synchronize
for gib in (0.25, 1, 4, 8):
    a = np.arange(int(gib * (1 << 30)) // 4, dtype=np.float32)
    d = DeviceArray(a.shape, np.float32)
    for rep in range(2):
        synchronize(); t0 = time.perf_counter(); d.upload(a); synchronize(); t = time.perf_counter() - t0
        print("%.2f GiB upload pass %d: %.1f ms  %.1f GB/s" % (gib, rep, t * 1e3, a.nbytes / t / 1e9), flush=True)
    for rep in range(2):
        t0 = time.perf_counter(); b = d.get(); t = time.perf_counter() - t0
        print("%.2f GiB download pass %d: %.1f ms  %.1f GB/s   equal: %s" % (gib, rep, t * 1e3, a.nbytes / t / 1e9, bool(np.array_equal(a, b))), flush=True)
        del b
