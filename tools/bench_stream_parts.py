"""where the FITS -> operator -> FITS time goes: the four source / sink combinations of the out-of-core cube -> cube path"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spectral_cube_amd import SpectralCube, Gaussian1DKernel, io_fits, streaming
from spectral_cube_amd.device import synchronize
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
ny = nx = 1024
nz = int(gib * 2**30 / (ny * nx * 4)) // 8 * 8
shape = (nz, ny, nx)
rng = np.random.default_rng(0)
d8 = (rng.standard_normal((8, ny, nx)) + 1.0).astype(np.float32)
hdr = {"CTYPE1": "RA---TAN", "CTYPE2": "DEC--TAN", "CTYPE3": "VRAD", "CDELT1": -1e-4, "CDELT2": 1e-4, "CDELT3": 500.0, "CUNIT3": "m/s",
       "CRPIX1": 1.0, "CRPIX2": 1.0, "CRPIX3": 1.0, "CRVAL1": 10.0, "CRVAL2": 20.0, "CRVAL3": 0.0, "BUNIT": "K"}
tmp = tempfile.mkdtemp(dir=os.environ.get("SPC_BENCH_DIR", "/tmp"))
path = os.path.join(tmp, "cube.fits")
host = np.tile(d8, (nz // 8, 1, 1))
nbytes = host.nbytes
os.environ["SPC_HBM_BUDGET"] = str(1 << 40)
SpectralCube.read(host[:8], hdr)          # (warm the library)
with open(path, "wb") as f:
    cards = [io_fits._card("SIMPLE", True), io_fits._card("BITPIX", -32), io_fits._card("NAXIS", 3), io_fits._card("NAXIS1", nx),
             io_fits._card("NAXIS2", ny), io_fits._card("NAXIS3", nz)] + [io_fits._card(k, v) for k, v in hdr.items()] + ["END".ljust(80)]
    h = "".join(cards)
    f.write((h + " " * ((-len(h)) % 2880)).encode("ascii"))
    blk = d8.astype(">f4").tobytes()
    for i in range(nz // 8):
        f.write(blk)
    f.write(b"\0" * ((-f.tell()) % 2880))
os.environ["SPC_HBM_BUDGET"] = str(nbytes // 4)
big = SpectralCube.read(path)
arr = SpectralCube.read(host, hdr)
k = Gaussian1DKernel(4)
host_out = np.empty(shape, np.float32); host_out[:] = 0
def run(label, fn, reps=3):
    ts = []
    for _ in range(reps):
        synchronize(); t0 = time.perf_counter(); fn(); synchronize(); ts.append(time.perf_counter() - t0)
    print("%-44s best %7.1f ms = %5.1f GB/s each way   all: %s" % (label, min(ts) * 1e3, nbytes / min(ts) / 1e9, " ".join("%.0f" % (t * 1e3) for t in ts)), flush=True)
outp = os.path.join(tmp, "out.fits")
run("host array -> smooth -> host array", lambda: arr.spectral_smooth(k).stream_into(host_out))
run("FITS       -> smooth -> host array", lambda: big.spectral_smooth(k).stream_into(host_out))
run("host array -> smooth -> FITS (new file each)", lambda: (os.path.exists(outp) and os.remove(outp), arr.spectral_smooth(k).write(outp)))
run("FITS       -> smooth -> FITS (new file each)", lambda: (os.path.exists(outp) and os.remove(outp), big.spectral_smooth(k).write(outp)))
t0 = time.perf_counter(); os.remove(outp); print("removing the 8 GiB result: %.0f ms" % ((time.perf_counter() - t0) * 1e3))
