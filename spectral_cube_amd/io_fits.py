"""FITS cube reader feeding HBM through pinned staging buffers.

Mirrors the part of ``spectral_cube/io/fits.py`` (read_data_fits :63-172, load_fits_cube
:171-260) that matters for the hot path: primary-HDU (or chosen image-HDU) cubes with 3 axes, or
4 axes with a degenerate Stokes axis, any BITPIX, BSCALE/BZERO/BLANK, ``BUNIT`` -> meta, and the
``LazyMask(np.isfinite)`` the reference attaches.  No astropy: the header is parsed here, and the
payload never passes through numpy arithmetic -

    file --os.preadv (reader threads, GIL released)--> pinned host buffers
         --spc_memcpy_h2d (async, copy stream)-------> raw staging in HBM
         --spc_fits_to_f32 (byte swap + BSCALE/BZERO/BLANK at HBM speed)--> (nz, ny, nx) float32

so the host only moves bytes and the PCIe link is the limit.  There is no CPU fallback: without
the HIP library / a GPU ``load_cube`` raises.
"""
import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _lib
from .device import DeviceArray, Event, Stream
from .wcs import parse_header

BLOCK = 2880
_BYTES = {8: 1, 16: 2, 32: 4, 64: 8, -32: 4, -64: 8}


class FITSReadError(Exception):
    """spectral_cube.io.fits.FITSReadError"""


class FitsImage:
    """Location and encoding of one image HDU inside a FITS file."""

    def __init__(self, path, header, data_offset):
        self.path, self.header, self.data_offset = path, header, data_offset
        self.bitpix = int(header.get("BITPIX", 0))
        naxis = int(header.get("NAXIS", 0))
        self.axes = [int(header["NAXIS%d" % (i + 1)]) for i in range(naxis)]       # FITS order: x fastest
        self.bscale = float(header.get("BSCALE", 1.0))
        self.bzero = float(header.get("BZERO", 0.0))
        self.blank = header.get("BLANK", None)
        if self.bitpix not in _BYTES and naxis:
            raise FITSReadError("unsupported BITPIX %r" % self.bitpix)

    @property
    def nsamples(self):
        return int(np.prod(self.axes, dtype=np.int64)) if self.axes else 0

    @property
    def nbytes(self):
        return self.nsamples * _BYTES.get(self.bitpix, 0)


def _read_header_block(f):
    cards = []
    while True:
        blk = f.read(BLOCK)
        if len(blk) < BLOCK:
            raise FITSReadError("truncated FITS header")
        text = blk.decode("ascii", "replace")
        done = False
        for i in range(0, BLOCK, 80):
            card = text[i:i + 80]
            if card.startswith("END") and card[3:].strip() == "":
                done = True
                break
            cards.append(card)
        if done:
            return cards


def scan_hdus(path):
    """[(FitsImage)] for every HDU of the file (headers only; payloads are skipped)."""
    out = []
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        first = True
        while f.tell() < size:
            start = f.tell()
            head = f.read(8)
            f.seek(start)
            if first and head != b"SIMPLE  ":
                raise FITSReadError("%s is not a FITS file" % path)
            if not first and head != b"XTENSION":
                break
            cards = _read_header_block(f)
            hdr = parse_header("".join(cards))
            img = FitsImage(path, hdr, f.tell())
            nbytes = img.nbytes if (first or str(hdr.get("XTENSION", "")).strip() == "IMAGE") else \
                int(abs(int(hdr.get("BITPIX", 8))) // 8 * int(np.prod([int(hdr.get("NAXIS%d" % (i + 1), 0))
                    for i in range(int(hdr.get("NAXIS", 0)))], dtype=np.int64)) + int(hdr.get("PCOUNT", 0)))
            img.is_image = first or str(hdr.get("XTENSION", "")).strip() == "IMAGE"
            out.append(img)
            f.seek(img.data_offset + ((nbytes + BLOCK - 1) // BLOCK) * BLOCK)
            first = False
    return out


def find_image(path, hdu=None):
    """the HDU ``read_data_fits`` would pick: *hdu* if given, else the first HDU with data."""
    hdus = scan_hdus(path)
    if hdu is not None:
        if not (0 <= hdu < len(hdus)) or not hdus[hdu].is_image or hdus[hdu].nsamples == 0:
            raise FITSReadError("No data found in HDU {0}. You can try using the hdu= keyword argument "
                                "to read data from another HDU.".format(hdu))
        return hdus[hdu]
    for h in hdus:
        if h.is_image and h.nsamples:
            return h
    raise ValueError("No arrays found")


def cube_shape(img):
    """(nz, ny, nx) of a 3-axis image, or of a 4-axis image whose extra axis is degenerate
    (load_fits_cube accepts naxis 3 and 4; multi-Stokes files are outside this path)."""
    ax = list(img.axes)
    if len(ax) == 4:
        if ax[3] == 1:
            ax = ax[:3]
        elif ax[2] == 1:
            ax = [ax[0], ax[1], ax[3]]
        else:
            raise NotImplementedError("multi-Stokes cubes (StokesSpectralCube) are outside the accelerated path")
    if len(ax) != 3:
        raise FITSReadError("Data should be 3- or 4-dimensional")
    return ax[2], ax[1], ax[0]


def cube_header(img):
    """header with the degenerate 4th axis removed (what _split_stokes/_orient leave behind)."""
    h = dict(img.header)
    if len(img.axes) == 4:
        drop = 4 if img.axes[3] == 1 else 3
        for k in list(h):
            if k[-1:] == str(drop) and k[:-1] in ("NAXIS", "CTYPE", "CRVAL", "CRPIX", "CDELT", "CUNIT", "CROTA"):
                del h[k]
            elif k.startswith("PC") and "_" in k and str(drop) in k[2:].split("_"):
                del h[k]
        if drop == 3:                              # (x, y, stokes, spectral): spectral axis becomes axis 3
            for base in ("CTYPE", "CRVAL", "CRPIX", "CDELT", "CUNIT"):
                if base + "4" in img.header:
                    h[base + "3"] = img.header[base + "4"]
                    h.pop(base + "4", None)
        h["NAXIS"] = 3
        h["WCSAXES"] = 3
    return h


class _Pinned:
    def __init__(self, nbytes):
        p = C.c_void_p()
        _lib.call("spc_host_alloc", C.c_size_t(nbytes), C.byref(p))
        self.ptr, self.nbytes = p.value, nbytes
        self.view = (C.c_uint8 * nbytes).from_address(self.ptr)
        self.free_evt = None

    def close(self):
        if self.ptr:
            _lib.call("spc_host_free", C.c_void_p(self.ptr))
            self.ptr = 0


class Staging:
    """pinned staging buffers + their raw device twins + the copy stream of load_cube, kept across calls (the
    out-of-core strip loop of streaming.py calls load_cube once per strip: pinning 1 GiB of host memory per call
    would cost more than the strip's transfer)"""

    def __init__(self, device=0, chunk_bytes=128 << 20, nbuffers=8):
        self.device, self.chunk_bytes, self.nbuffers = device, int(chunk_bytes), int(nbuffers)
        self.pinned = [_Pinned(self.chunk_bytes) for _ in range(self.nbuffers)]
        self.d_raw = [DeviceArray((self.chunk_bytes,), np.uint8, device) for _ in range(self.nbuffers)]
        self.stream = Stream(device)

    def close(self):
        for b in self.pinned:
            b.close()
        self.pinned, self.d_raw = [], []


def load_cube(path, device=0, hdu=None, chunk_bytes=128 << 20, nbuffers=8, readers=8, stats=None, rows=None,
              staging=None, dtype=np.float32):
    """Stream the image payload of *path* into a (nz, ny, nx) float32 DeviceArray (``dtype=np.float64``: a BITPIX = -64 / 32 /
    64 image in its own precision, spc_fits_to_f64 - what astropy hands the reference, io/fits.py:63-172).

    chunk_bytes / nbuffers: size and count of the pinned staging buffers; readers: threads
    filling them with os.preadv; rows=(y0, y1): load only that row strip of every plane (the
    (y, x)-tile sharding of SURVEY.md section 8e; CRPIX2 of the returned header is shifted).
    Returns (DeviceArray, header dict).  *stats*, if a dict,
    receives {"bytes", "seconds"} of the payload transfer (the PCIe-inclusive rate).
    Measured on the MI355X box, 4 GiB float32 file in the page cache (profiles/r01_fits_reader.log):
    1 reader 7.9 GB/s, 4 readers 24 GB/s, 8 readers x 128 MiB buffers 44 GB/s (the PCIe Gen5 x16
    link), i.e. 1.1e4 Mvoxel/s end to end - two orders below the HBM-resident kernels."""
    import time
    _lib.require_gpu()
    img = find_image(path, hdu)
    nz, ny_file, nx = cube_shape(img)
    y0, y1 = (0, ny_file) if rows is None else (int(rows[0]), int(rows[1]))
    if not (0 <= y0 < y1 <= ny_file):
        raise ValueError("rows must satisfy 0 <= y0 < y1 <= %d" % ny_file)
    ny = y1 - y0
    bps = _BYTES[img.bitpix]
    if staging is not None:
        chunk_bytes, nbuffers = staging.chunk_bytes, staging.nbuffers
    wide = np.dtype(dtype) == np.float64
    if wide and img.bitpix not in (-64, 32, 64):
        raise ValueError("dtype=float64 is for BITPIX = -64 / 32 / 64 images (this one: %d)" % img.bitpix)
    out = DeviceArray((nz, ny, nx), np.float64 if wide else np.float32, device)
    seg = ny * nx * bps                                # one plane's strip: contiguous in the file
    total = nz * seg
    if rows is None:                                   # whole planes: the payload is ONE contiguous run
        chunk = max(bps * 4, min(chunk_bytes, total) // (bps * 4) * (bps * 4))      # whole samples, 16-byte friendly
        nchunks = (total + chunk - 1) // chunk
        ppc = 0
    else:                                              # row strip (multi-GPU sharding): whole strips per chunk
        ppc = max(1, min(nz, chunk_bytes // seg))      # planes per chunk
        chunk = ppc * seg
        nchunks = (nz + ppc - 1) // ppc
    nbuf = max(1, min(nbuffers, nchunks))
    if staging is not None:
        if chunk > staging.chunk_bytes:
            raise ValueError("one plane strip (%d bytes) does not fit the staging buffers (%d bytes)" % (chunk, staging.chunk_bytes))
        pinned, d_raw, stream = staging.pinned[:nbuf], staging.d_raw[:nbuf], staging.stream
        for b in pinned:
            b.free_evt = None
    else:
        pinned = [_Pinned(chunk) for _ in range(nbuf)]
        d_raw = [DeviceArray((chunk,), np.uint8, device) for _ in range(nbuf)]
        stream = Stream(device)
    fd = os.open(path, os.O_RDONLY)
    has_blank = img.blank is not None and img.bitpix > 0
    t0 = time.perf_counter()
    try:
        def pread_full(mv, offset):
            got, n = 0, len(mv)
            while got < n:
                r = os.preadv(fd, [mv[got:]], offset + got)
                if r <= 0:
                    raise FITSReadError("truncated FITS payload")
                got += r

        def fill(i):
            b = pinned[i % nbuf]
            if rows is None:
                off, n = i * chunk, min(chunk, total - i * chunk)
                pread_full(memoryview(b.view)[:n], img.data_offset + off)
                return i, n
            z0, z1 = i * ppc, min(nz, (i + 1) * ppc)
            mv = memoryview(b.view)
            for k, z in enumerate(range(z0, z1)):
                pread_full(mv[k * seg:(k + 1) * seg], img.data_offset + (z * ny_file + y0) * nx * bps)
            return i, (z1 - z0) * seg

        with ThreadPoolExecutor(max_workers=max(1, readers)) as pool:
            pending = {}
            nxt = 0

            def submit(i):
                b = pinned[i % nbuf]
                if b.free_evt is not None:          # the H2D that last used this buffer must be done
                    b.free_evt.synchronize()
                    b.free_evt = None
                pending[i] = pool.submit(fill, i)

            while nxt < min(nbuf, nchunks):
                submit(nxt)
                nxt += 1
            for i in range(nchunks):
                _, n = pending.pop(i).result()
                b, raw = pinned[i % nbuf], d_raw[i % nbuf]
                _lib.call("spc_memcpy_h2d", device, C.c_void_p(raw.ptr), C.c_void_p(b.ptr), C.c_size_t(n), stream.handle)
                ev = Event(device)
                ev.record(stream)
                b.free_evt = ev
                first = (i * chunk) // bps
                _lib.call("spc_fits_to_f64" if wide else "spc_fits_to_f32", device, stream.handle, C.c_void_p(raw.ptr), img.bitpix,
                          img.bscale, img.bzero, 1 if has_blank else 0, int(img.blank) if has_blank else 0,
                          n // bps, C.c_void_p(out.ptr + first * (8 if wide else 4)))
                if nxt < nchunks:
                    submit(nxt)
                    nxt += 1
        stream.synchronize()
    finally:
        os.close(fd)
        if staging is None:
            for b in pinned:
                b.close()
    if stats is not None:
        stats.update(bytes=total, seconds=time.perf_counter() - t0)
    if staging is None:
        out._keep = d_raw
    hdr = cube_header(img)
    if rows is not None:
        hdr["NAXIS2"] = ny
        if "CRPIX2" in hdr:
            hdr["CRPIX2"] = float(hdr["CRPIX2"]) - y0
    return out, hdr


# ---- writer (tests / fixtures; the reference writes through astropy, io/fits.py:262-294) ----------
def _card(key, value, comment=""):
    if isinstance(value, bool):
        v = "%20s" % ("T" if value else "F")
    elif isinstance(value, (int, np.integer)):
        v = "%20d" % int(value)
    elif isinstance(value, (float, np.floating)):
        v = "%20s" % repr(float(value)).upper().replace("E+", "E")
    else:
        v = "'%-8s'" % str(value).replace("'", "''")
    card = "%-8s= %s" % (key[:8], v)
    if comment:
        card += " / " + comment
    return card[:80].ljust(80)


def write_fits(path, data, header=None, bitpix=None, bscale=None, bzero=None, blank=None):
    """Minimal FITS writer: *data* (any numpy array, written big-endian as it is) plus header keys."""
    data = np.asarray(data)
    if bitpix is None:
        bitpix = {np.dtype("f4"): -32, np.dtype("f8"): -64, np.dtype("i2"): 16, np.dtype("i4"): 32,
                  np.dtype("u1"): 8, np.dtype("i8"): 64}[data.dtype.newbyteorder("=")]
    cards = [_card("SIMPLE", True), _card("BITPIX", bitpix), _card("NAXIS", data.ndim)]
    for i, n in enumerate(data.shape[::-1]):
        cards.append(_card("NAXIS%d" % (i + 1), n))
    if bscale is not None:
        cards.append(_card("BSCALE", float(bscale)))
    if bzero is not None:
        cards.append(_card("BZERO", float(bzero)))
    if blank is not None:
        cards.append(_card("BLANK", int(blank)))
    for k, v in (parse_header(header) if header is not None else {}).items():
        if k in ("SIMPLE", "BITPIX", "BSCALE", "BZERO", "BLANK", "EXTEND") or k.startswith("NAXIS"):
            continue
        cards.append(_card(k, v))
    cards.append("END".ljust(80))
    text = "".join(cards)
    text += " " * ((-len(text)) % BLOCK)
    payload = data.astype(data.dtype.newbyteorder(">")).tobytes()
    part = "%s.part-%d" % (path, os.getpid())              # a failed write never leaves a truncated target (like FitsSink)
    try:
        with open(part, "wb") as f:
            f.write(text.encode("ascii"))
            f.write(payload)
            f.write(b"\0" * ((-len(payload)) % BLOCK))
        os.replace(part, path)
    except BaseException:
        try:
            os.unlink(part)
        except OSError:
            pass
        raise


def save_cube(path, dev, header=None, chunk_bytes=128 << 20, nbuffers=4, overwrite=False):
    """Write a (nz, ny, nx) float32 DeviceArray as a BITPIX=-32 FITS file (the reference's
    write_fits_cube, spectral_cube/io/fits.py:262-294, without the history cards): the byte swap
    runs on the device (spc_fits_to_f32 with BITPIX -32 is its own inverse), chunks come back
    through pinned buffers and are written with os.pwrite while the next chunk is in flight."""
    if os.path.exists(path) and not overwrite:
        raise OSError("File %r already exists (use overwrite=True)" % path)
    if dev.dtype != np.float32 or len(dev.shape) != 3 or getattr(dev, "_is_view", False):
        raise TypeError("save_cube needs a contiguous float32 (nz, ny, nx) DeviceArray")
    device = dev.device
    nz, ny, nx = dev.shape
    hdr = parse_header(header) if header is not None else {}
    cards = [_card("SIMPLE", True), _card("BITPIX", -32), _card("NAXIS", 3),
             _card("NAXIS1", nx), _card("NAXIS2", ny), _card("NAXIS3", nz)]
    for k, v in hdr.items():
        if k in ("SIMPLE", "BITPIX", "BSCALE", "BZERO", "BLANK", "EXTEND", "WCSAXES") or k.startswith("NAXIS"):
            continue
        cards.append(_card(k, v))
    cards.append("END".ljust(80))
    text = "".join(cards)
    text += " " * ((-len(text)) % BLOCK)
    total = nz * ny * nx * 4
    chunk = max(16, min(chunk_bytes, total) // 16 * 16)
    nchunks = (total + chunk - 1) // chunk
    nbuf = max(1, min(nbuffers, nchunks))
    pinned = [_Pinned(chunk) for _ in range(nbuf)]
    d_swap = [DeviceArray((chunk,), np.uint8, device) for _ in range(nbuf)]
    stream = Stream(device)
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    try:
        os.pwrite(fd, text.encode("ascii"), 0)
        base = len(text)
        events = [None] * nbuf

        def flush(i):
            b = pinned[i % nbuf]
            events[i % nbuf].synchronize()
            n = min(chunk, total - i * chunk)
            mv, done = memoryview(b.view)[:n], 0
            while done < n:
                done += os.pwrite(fd, mv[done:], base + i * chunk + done)

        for i in range(nchunks):
            if i >= nbuf:
                flush(i - nbuf)
            n = min(chunk, total - i * chunk)
            _lib.call("spc_fits_to_f32", device, stream.handle, C.c_void_p(dev.ptr + i * chunk), -32, 1.0, 0.0, 0, 0,
                      n // 4, C.c_void_p(d_swap[i % nbuf].ptr))
            _lib.call("spc_memcpy_d2h", device, C.c_void_p(pinned[i % nbuf].ptr), C.c_void_p(d_swap[i % nbuf].ptr),
                      C.c_size_t(n), stream.handle)
            ev = Event(device)
            ev.record(stream)
            events[i % nbuf] = ev
        for i in range(max(0, nchunks - nbuf), nchunks):
            flush(i)
        pad = (-total) % BLOCK
        if pad:
            os.pwrite(fd, b"\0" * pad, base + total)
    finally:
        os.close(fd)
        for b in pinned:
            b.close()


# ---- CASA-style BEAMS binary table (per-channel beams of a VaryingResolutionSpectralCube) ----------
_TFORM = {"E": ">f4", "D": ">f8", "J": ">i4", "I": ">i2", "K": ">i8", "B": "u1"}
_UNIT_DEG = {"arcsec": 1.0 / 3600.0, "arcmin": 1.0 / 60.0, "deg": 1.0, "degree": 1.0, "degrees": 1.0,
             "rad": 180.0 / np.pi, "mas": 1.0 / 3.6e6}


def read_beams_table(path):
    """The ``BEAMS`` BINTABLE extension of *path* (read_data_fits, spectral_cube/io/fits.py:94-131),
    or None when the file has none: {"BMAJ", "BMIN" (degrees), "BPA" (degrees), "CHAN", "POL"} as
    float64 / int arrays.  Units come from the TUNITn of the BMAJ / BMIN columns (arcsec when absent,
    AIPS's 'DEGREES' accepted); BPA is taken as degrees, as the reference does."""
    for h in scan_hdus(path):
        hdr = h.header
        if str(hdr.get("XTENSION", "")).strip() != "BINTABLE" or str(hdr.get("EXTNAME", "")).strip() != "BEAMS":
            continue
        nrow, rowlen, nf = int(hdr["NAXIS2"]), int(hdr["NAXIS1"]), int(hdr["TFIELDS"])
        fields, units = [], {}
        for i in range(1, nf + 1):
            form = str(hdr["TFORM%d" % i]).strip()
            rep, code = (int(form[:-1]) if form[:-1] else 1), form[-1]
            if code not in _TFORM or rep != 1:
                raise FITSReadError("unsupported BEAMS column format %r" % form)
            name = str(hdr.get("TTYPE%d" % i, "COL%d" % i)).strip()
            fields.append((name, _TFORM[code]))
            units[name] = str(hdr.get("TUNIT%d" % i, "arcsec")).strip()
        dt = np.dtype(fields)
        if dt.itemsize != rowlen:
            raise FITSReadError("BEAMS table row length %d does not match its columns (%d)" % (rowlen, dt.itemsize))
        with open(path, "rb") as f:
            f.seek(h.data_offset)
            rec = np.frombuffer(f.read(nrow * rowlen), dtype=dt, count=nrow)
        out = {}
        for name in dt.names:
            col = rec[name]
            if name in ("BMAJ", "BMIN"):
                u = units[name].lower()
                if u not in _UNIT_DEG:
                    raise FITSReadError("unsupported beam unit %r" % units[name])
                out[name] = col.astype(np.float64) * _UNIT_DEG[u]
            elif name == "BPA":
                out[name] = col.astype(np.float64)
            else:
                out[name] = col.astype(np.int64)
        for need in ("BMAJ", "BMIN", "BPA"):
            if need not in out:
                raise FITSReadError("BEAMS table lacks a %s column" % need)
        return out
    return None


def append_beams_table(path, bmaj_deg, bmin_deg, bpa_deg, chan=None, pol=None):
    """Append a ``BEAMS`` BINTABLE HDU (arcsec / arcsec / deg, float32, + CHAN, POL int32) to an
    existing FITS file: what beams_to_bintable / VaryingResolutionSpectralCube.hdulist produce
    (dask_spectral_cube.py:1493-1509)."""
    n = len(bmaj_deg)
    rec = np.zeros(n, dtype=[("BMAJ", ">f4"), ("BMIN", ">f4"), ("BPA", ">f4"), ("CHAN", ">i4"), ("POL", ">i4")])
    rec["BMAJ"] = np.asarray(bmaj_deg, dtype=np.float64) * 3600.0
    rec["BMIN"] = np.asarray(bmin_deg, dtype=np.float64) * 3600.0
    rec["BPA"] = bpa_deg
    rec["CHAN"] = np.arange(n) if chan is None else chan
    if pol is not None:
        rec["POL"] = pol
    cards = [_card("XTENSION", "BINTABLE"), _card("BITPIX", 8), _card("NAXIS", 2), _card("NAXIS1", rec.dtype.itemsize),
             _card("NAXIS2", n), _card("PCOUNT", 0), _card("GCOUNT", 1), _card("TFIELDS", 5)]
    for i, (name, form, unit) in enumerate((("BMAJ", "1E", "arcsec"), ("BMIN", "1E", "arcsec"), ("BPA", "1E", "deg"),
                                            ("CHAN", "1J", None), ("POL", "1J", None)), 1):
        cards += [_card("TTYPE%d" % i, name), _card("TFORM%d" % i, form)]
        if unit:
            cards.append(_card("TUNIT%d" % i, unit))
    cards += [_card("EXTNAME", "BEAMS"), _card("EXTVER", 1), _card("NCHAN", n), _card("NPOL", 1), "END".ljust(80)]
    text = "".join(cards)
    text += " " * ((-len(text)) % BLOCK)
    payload = rec.tobytes()
    with open(path, "ab") as f:
        if f.tell() % BLOCK:
            raise FITSReadError("%s is not padded to a FITS block" % path)
        f.write(text.encode("ascii"))
        f.write(payload)
        f.write(b"\0" * ((-len(payload)) % BLOCK))
