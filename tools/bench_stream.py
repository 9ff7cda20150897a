"""Scratch/report: PCIe-inclusive rate of the OUT-OF-CORE path (streaming.py): moment 0 + 1 + 2 of a cube
four times the HBM budget, from a FITS file in the page cache and from a host array, row strips staged by a
worker thread under the kernels of the previous strip.  Prints wall clock, GB/s and Mvoxel/s next to the
resident kernel's figure."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spectral_cube_amd import SpectralCube, io_fits, streaming
from spectral_cube_amd.device import synchronize

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
ny = nx = 1024
nz = int(gib * 2**30 / (ny * nx * 4)) // 8 * 8
shape = (nz, ny, nx)
rng = np.random.default_rng(0)
d8 = (rng.standard_normal((8, ny, nx)) + 1.0).astype(np.float32)
hdr = {"CTYPE1": "RA---TAN", "CTYPE2": "DEC--TAN", "CTYPE3": "VRAD", "CDELT1": -1e-4, "CDELT2": 1e-4, "CDELT3": 500.0, "CUNIT3": "m/s",
       "CRPIX1": 1.0, "CRPIX2": 1.0, "CRPIX3": 1.0, "CRVAL1": 10.0, "CRVAL2": 20.0, "CRVAL3": 0.0, "BUNIT": "K"}
# SPC_BENCH_DIR: where the FITS files live (default /dev/shm; tmpfs on the GPU box takes os.pwrite at 7.5 GB/s into a fresh file,
# its /tmp - a disk-backed file system, page cache - at 15 - 20 GB/s: tools/bench_filewrite.py)
_base = os.environ.get("SPC_BENCH_DIR", "/dev/shm" if os.path.isdir("/dev/shm") else None)
tmp = tempfile.mkdtemp(dir=_base)
path = os.path.join(tmp, "cube.fits")
with open(path, "wb") as f:
    cards = [io_fits._card("SIMPLE", True), io_fits._card("BITPIX", -32), io_fits._card("NAXIS", 3), io_fits._card("NAXIS1", nx),
             io_fits._card("NAXIS2", ny), io_fits._card("NAXIS3", nz)] + [io_fits._card(k, v) for k, v in hdr.items()] + ["END".ljust(80)]
    h = "".join(cards)
    f.write((h + " " * ((-len(h)) % 2880)).encode("ascii"))
    blk = d8.astype(">f4").tobytes()
    for i in range(nz // 8):
        f.write(blk)
    f.write(b"\0" * ((-f.tell()) % 2880))
nbytes = nz * ny * nx * 4
print("cube %s = %.1f GiB, budget = 1/4 of it" % (shape, nbytes / 2**30), flush=True)
os.environ["SPC_HBM_BUDGET"] = str(nbytes // 4)


def timed(cube, label, reps=5):
    ts = []
    for _ in range(reps):
        synchronize(); t0 = time.perf_counter()
        m = cube.moments012()
        synchronize(); ts.append(time.perf_counter() - t0)
    best = min(ts)
    print("%-58s %8.1f ms  %6.1f GB/s  %8.0f Mvoxel/s   all runs (ms): %s" % (
        label, best * 1e3, nbytes / best / 1e9, nz * ny * nx / best / 1e6, " ".join("%.0f" % (t * 1e3) for t in ts)), flush=True)
    return m


def read_only(src, label, rows):
    """the readers alone: every chunk of every strip into pinned memory, nothing sent to the device"""
    from concurrent.futures import ThreadPoolExecutor
    from spectral_cube_amd.io_fits import _Pinned
    nzs, nys, nxs = src.shape
    seg = rows * nxs * src.sample_bytes
    ppc = max(1, (32 << 20) // seg)
    bufs = [_Pinned(ppc * seg) for _ in range(16)]
    tasks = [(y0, min(nys, y0 + rows), z0, min(nzs, z0 + ppc)) for y0 in range(0, nys, rows) for z0 in range(0, nzs, ppc)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(8) as ex:
        list(ex.map(lambda it: src.read_into(bufs[it[0] % 16].view, it[1][2], it[1][3], it[1][0], it[1][1]), enumerate(tasks)))
    dt = time.perf_counter() - t0
    for b in bufs:
        b.close()
    print("%-58s %8.1f ms  %6.1f GB/s" % (label, dt * 1e3, nbytes / dt / 1e9), flush=True)


big = SpectralCube.read(path)
assert big._stream_source() is not None
rows = streaming.plan_rows(shape, streaming.hbm_budget(0))
read_only(big._stream_source(), "readers only: FITS preads -> pinned (8 threads)", rows)
m_f = timed(big, "FITS file (page cache) -> strips of %d rows -> moments012" % rows)
host = np.tile(d8, (nz // 8, 1, 1))
arr = SpectralCube.read(host, hdr)
assert arr._stream_source() is not None
read_only(arr._stream_source(), "readers only: ndarray rows -> pinned (8 threads)", rows)
m_a = timed(arr, "host float32 array (pageable) -> strips -> moments012")
# cube -> cube out of core: read -> spectral_smooth (33 taps) -> write, the result as large as the input
from spectral_cube_amd import Gaussian1DKernel
outp = os.path.join(tmp, "smoothed.fits")
ts = []
for _ in range(3):
    synchronize(); t0 = time.perf_counter()
    big.spectral_smooth(Gaussian1DKernel(4)).write(outp, overwrite=True)
    synchronize(); ts.append(time.perf_counter() - t0)
best = min(ts)
print("%-58s %8.1f ms  %6.1f GB/s in + %.1f GB/s out  %8.0f Mvoxel/s   all runs (ms): %s" % (
    "FITS -> spectral_smooth(33 taps) -> FITS, strips both ways", best * 1e3, nbytes / best / 1e9, nbytes / best / 1e9,
    nz * ny * nx / best / 1e6, " ".join("%.0f" % (t * 1e3) for t in ts)), flush=True)
host_out = np.empty(shape, np.float32)
ts = []
for _ in range(3):
    synchronize(); t0 = time.perf_counter()
    arr.spectral_smooth(Gaussian1DKernel(4)).stream_into(host_out)
    synchronize(); ts.append(time.perf_counter() - t0)
best = min(ts)
print("%-58s %8.1f ms  %6.1f GB/s in + %.1f GB/s out  %8.0f Mvoxel/s   all runs (ms): %s" % (
    "host array -> spectral_smooth(33 taps) -> host array", best * 1e3, nbytes / best / 1e9, nbytes / best / 1e9,
    nz * ny * nx / best / 1e6, " ".join("%.0f" % (t * 1e3) for t in ts)), flush=True)
# operators on whole planes: row strips with halo rows (spatial_smooth), slabs of channels (reproject, statistics along y)
from spectral_cube_amd import Gaussian2DKernel


def cube_to_cube(label, make, out_shape):
    dst = np.empty(out_shape, np.float32)
    ts = []
    for _ in range(3):
        synchronize(); t0 = time.perf_counter()
        make().stream_into(dst)
        synchronize(); ts.append(time.perf_counter() - t0)
    best = min(ts)
    print("%-58s %8.1f ms  %6.1f GB/s in + %.1f GB/s out  %8.0f Mvoxel/s   all runs (ms): %s" % (
        label, best * 1e3, nbytes / best / 1e9, dst.nbytes / best / 1e9, nz * ny * nx / best / 1e6, " ".join("%.0f" % (t * 1e3) for t in ts)), flush=True)


cube_to_cube("host array -> spatial_smooth(29x29), slabs of planes -> host array", lambda: arr.spatial_smooth(Gaussian2DKernel(8 / 2.35482)), shape)
c_, s_ = np.cos(np.radians(30)), np.sin(np.radians(30))
target = {k: v for k, v in hdr.items() if not k.endswith("3")}
target.update(NAXIS=2, NAXIS1=nx, NAXIS2=ny, CRPIX1=nx / 2.0, CRPIX2=ny / 2.0, CRVAL1=10.0 - 1e-4 * nx / 2, CRVAL2=20.0 + 1e-4 * ny / 2,
              PC1_1=c_, PC1_2=-s_, PC2_1=s_, PC2_2=c_)
arr.allow_huge_operations = True
cube_to_cube("host array -> reproject (30 deg), slabs of planes -> host array", lambda: arr.reproject(target), shape)
ts = []
for _ in range(3):
    synchronize(); t0 = time.perf_counter()
    arr.median(axis=1)
    synchronize(); ts.append(time.perf_counter() - t0)
best = min(ts)
print("%-58s %8.1f ms  %6.1f GB/s  %8.0f Mvoxel/s   all runs (ms): %s" % (
    "host array -> slabs of planes -> median along y", best * 1e3, nbytes / best / 1e9, nz * ny * nx / best / 1e6,
    " ".join("%.0f" % (t * 1e3) for t in ts)), flush=True)
os.remove(outp)
os.environ["SPC_HBM_BUDGET"] = str(1 << 42)
res = SpectralCube.read(path)
res._device_data(); synchronize()
m_r = timed(res, "resident cube (kernel + map download only)", reps=3)
for got in (m_f, m_a):
    for a, b in zip(got, m_r):
        a, b = np.asarray(a), np.asarray(b)
        assert np.array_equal(np.isnan(a), np.isnan(b)) and np.nanmax(np.abs(a - b)) <= 1e-12 * np.nanmax(np.abs(b))
print("streamed maps == resident maps (1e-12 of the map's range; the z split of the kernel follows the strip size)")
os.remove(path); os.rmdir(tmp)
