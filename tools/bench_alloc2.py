"""how fast can a fresh 4 GiB host result array be made writable? (first-touch page faults dominate the cube -> cube dask entry)"""
import ctypes, mmap, os, sys, time
import numpy as np
from concurrent.futures import ThreadPoolExecutor
libc = ctypes.CDLL(None, use_errno=True)
N = 4 << 30
def touch(a, nthreads):
    v = a.view(np.uint8).reshape(-1)
    step = len(v) // nthreads
    def f(i):
        v[i * step:(i + 1) * step:4096] = 1
    with ThreadPoolExecutor(nthreads) as ex:
        list(ex.map(f, range(nthreads)))
for label, adv, nt in (("plain, 1 thread", None, 1), ("plain, 8 threads", None, 8), ("plain, 32 threads", None, 32), ("MADV_HUGEPAGE, 8 threads", 14, 8), ("MADV_POPULATE_WRITE x 8 threads", 23, 8)):
    t0 = time.perf_counter()
    a = np.empty(N // 4, np.float32)
    addr = a.ctypes.data
    pa = (addr + 4095) & ~4095
    if adv == 14:
        r = libc.madvise(ctypes.c_void_p(pa), ctypes.c_size_t(N - 4096), 14)
        touch(a, nt)
    elif adv == 23:
        step = ((N - 8192) // nt) & ~4095
        def pop(i):
            return libc.madvise(ctypes.c_void_p(pa + i * step), ctypes.c_size_t(step), 23)
        with ThreadPoolExecutor(nt) as ex:
            r = list(ex.map(pop, range(nt)))
    else:
        r = None
        touch(a, nt)
    t = time.perf_counter() - t0
    print("%-36s %7.1f ms  %5.1f GB/s  rc=%s" % (label, t * 1e3, N / t / 1e9, r if not isinstance(r, list) else set(r)), flush=True)
    del a
print(open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), os.cpu_count())
