"""Scratch/report: PCIe-inclusive rate of the FITS reader (file in the page cache -> pinned
buffers -> HBM -> float32 decode), then moments of the loaded cube."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spectral_cube_amd import io_fits, ops, synth
from spectral_cube_amd.device import DeviceArray, synchronize

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
nz = int(gib * 2**30 / (1024 * 1024 * 4))
shape = (nz, 1024, 1024)
d = np.random.default_rng(0).standard_normal((8, 1024, 1024)).astype(np.float32)
tmp = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
path = os.path.join(tmp, "cube.fits")
t0 = time.perf_counter()
io_fits.write_fits(path, np.broadcast_to(d[:1], (1, 1024, 1024)), {"CTYPE3": "VRAD"})
# extend the payload to the full size (header says 1 plane; rewrite header + append planes)
with open(path, "wb") as f:
    hdr = "".join([io_fits._card("SIMPLE", True), io_fits._card("BITPIX", -32), io_fits._card("NAXIS", 3),
                   io_fits._card("NAXIS1", 1024), io_fits._card("NAXIS2", 1024), io_fits._card("NAXIS3", nz),
                   io_fits._card("CTYPE3", "VRAD"), "END".ljust(80)])
    f.write((hdr + " " * ((-len(hdr)) % 2880)).encode("ascii"))
    blk = d.astype(">f4").tobytes()
    for i in range(nz // 8):
        f.write(blk)
    f.write(b"\0" * ((-f.tell()) % 2880))
print("wrote %.1f GiB in %.1f s" % (os.path.getsize(path) / 2**30, time.perf_counter() - t0), flush=True)
for readers, nbuf, chunk in ((1, 2, 256), (4, 4, 256), (8, 8, 128), (16, 16, 64)):
    best = None
    for rep in range(2):
        st = {}
        dev, h = io_fits.load_cube(path, chunk_bytes=chunk << 20, nbuffers=nbuf, readers=readers, stats=st)
        synchronize()
        rate = st["bytes"] / st["seconds"] / 1e9
        best = max(best or 0, rate)
        del dev
    print("readers=%2d buffers=%2d chunk=%3d MiB : %6.1f GB/s  (%7.0f Mvoxel/s PCIe-inclusive)" % (readers, nbuf, chunk, best, best * 1e3 / 4), flush=True)
dev, h = io_fits.load_cube(path, readers=8, nbuffers=8, chunk_bytes=128 << 20)
got = dev.get()[:8]
assert np.array_equal(got, d), "decode mismatch"
print("decode check ok")
os.remove(path); os.rmdir(tmp)
