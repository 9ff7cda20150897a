"""what a fresh 4 GiB file costs on this box: 8 threads of os.pwrite (64 MiB pieces) into /dev/shm and /tmp, with and without
posix_fallocate first - the ceiling of the FITS sink of the out-of-core pipeline"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
N, P = 4 << 30, 64 << 20
buf = bytes(P)
for d in ("/dev/shm", "/tmp"):
    for falloc in (False, True):
        path = os.path.join(d, "spc_wtest.bin")
        t0 = time.perf_counter()
        fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        ta = 0.0
        if falloc:
            os.posix_fallocate(fd, 0, N); ta = time.perf_counter() - t0
        else:
            os.ftruncate(fd, N)
        with ThreadPoolExecutor(8) as ex:
            list(ex.map(lambda off: os.pwrite(fd, buf, off), range(0, N, P)))
        os.close(fd)
        t = time.perf_counter() - t0
        t1 = time.perf_counter()
        fd = os.open(path, os.O_WRONLY)
        with ThreadPoolExecutor(8) as ex:
            list(ex.map(lambda off: os.pwrite(fd, buf, off), range(0, N, P)))
        os.close(fd)
        t2 = time.perf_counter() - t1
        os.unlink(path)
        print("%-9s fallocate=%-5s fresh file %7.1f ms = %5.1f GB/s (fallocate itself %6.1f ms)   rewrite in place %7.1f ms = %5.1f GB/s" % (
            d, falloc, t * 1e3, N / t / 1e9, ta * 1e3, t2 * 1e3, N / t2 / 1e9), flush=True)
