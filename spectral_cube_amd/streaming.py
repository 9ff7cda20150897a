"""Out-of-core path: cubes larger than the HBM budget, processed in (y, x) row strips.

The reference handles cubes larger than memory by walking them slice by slice or ray by ray
(``_moments.py:89-168``), by its huge-cube strategy switch (``cube_utils.py:266-301``) and, in the Dask
class, by rechunking to ``(-1, 'auto', 'auto')`` (``dask_spectral_cube.py:551,618``): spectral axis whole,
the image plane cut up.  Here the cube stays where it is - a FITS file, a memory map, a host array - and
goes through HBM as row strips ``(nz, rows, nx)``: x contiguity kept (the kernels' coalescing), every
spaxel whole (moments, argmax, the spectral stencil and statistics need no halo).  A worker thread stages
strip k + 1 (file -> pinned buffers -> H2D -> device decode for FITS; strided H2D for arrays) while the
kernels of strip k run on their own stream; the 2-D maps are assembled on the device - a strip's kernel
writes rows [y0, y1) of the final map in place.

What streams: moment 0 / 1 / 2, argmax / argmin / max / min along the spectral axis and of the whole cube,
spectral_smooth(...).moment (the fused kernels), statistics() and the axis=None reductions.  Everything else
asks for the resident cube and raises HugeCubeError with the budget in the message.

SPC_HBM_BUDGET (bytes; K / M / G suffixes) overrides the default budget = 80 % of the free HBM.
"""
import ctypes as C
import os
import threading

import numpy as np

from . import _lib
from .device import DeviceArray, Stream


class HugeCubeError(MemoryError):
    """the operation needs the whole cube resident in HBM and the cube is larger than the budget"""


_SUFFIX = {"K": 1 << 10, "M": 1 << 20, "G": 1 << 30, "T": 1 << 40}


def parse_bytes(text):
    t = str(text).strip().upper().rstrip("B")
    if t and t[-1] == "I":
        t = t[:-1]
    mult = 1
    if t and t[-1] in _SUFFIX:
        mult, t = _SUFFIX[t[-1]], t[:-1]
    return int(float(t) * mult)


def hbm_budget(device=0):
    """bytes a cube (data + mask array) may take to be made resident"""
    env = os.environ.get("SPC_HBM_BUDGET")
    if env:
        return parse_bytes(env)
    from .device import device_info
    return int(0.8 * device_info(device)["free_mem"])


def _env_int(name, default):
    try:
        return int(os.environ.get(name, default))
    except ValueError:
        return default


# ---- sources: where the strips come from -------------------------------------------------------------------
class NdarraySource:
    """host array / numpy memmap, (nz, ny, nx), any real dtype (also the uint8 array term of a mask).  Reader
    threads copy (and convert) plane ranges of a strip into pinned staging buffers - numpy releases the GIL for
    these copies - from where they go up with asynchronous copies at the link rate; one synchronous copy from
    pageable memory is staged by the runtime on a single thread (16 GB/s measured, whatever the number of callers)."""

    def __init__(self, data, out_dtype=np.float32):
        self.data = data
        self.shape = tuple(int(s) for s in data.shape)
        self.out_dtype = np.dtype(out_dtype)
        self.sample_bytes = self.out_dtype.itemsize          # bytes per sample in the staging buffer
        self.decode = None                                   # staged bytes ARE the device representation

    def read_into(self, view_u8, z0, z1, y0, y1):
        nz, ny, nx = self.shape
        n = (z1 - z0) * (y1 - y0) * nx
        dst = np.frombuffer(view_u8, dtype=self.out_dtype, count=n).reshape(z1 - z0, y1 - y0, nx)
        src = self.data[z0:z1, y0:y1]
        if self.out_dtype == np.uint8 and src.dtype == np.bool_:
            src = src.view(np.uint8)
        np.copyto(dst, src, casting="unsafe")
        return n * self.sample_bytes

    def release(self):
        pass


class FitsSource:
    """image HDU of a FITS file: reader threads pread the big-endian plane strips into the pinned buffers, the
    device decodes them (byte swap / BSCALE / BZERO / BLANK: spc_fits_to_f32)"""

    def __init__(self, path, hdu=None):
        from . import io_fits
        self.path, self.hdu = os.fspath(path), hdu
        self.img = io_fits.find_image(self.path, hdu)
        self.shape = tuple(io_fits.cube_shape(self.img))
        self.out_dtype = np.dtype(np.float32)
        self.sample_bytes = io_fits._BYTES[self.img.bitpix]
        self._fd = None
        self._lock = threading.Lock()

    def _file(self):
        with self._lock:
            if self._fd is None:
                self._fd = os.open(self.path, os.O_RDONLY)
        return self._fd

    def read_into(self, view_u8, z0, z1, y0, y1):
        from .io_fits import FITSReadError
        nz, ny, nx = self.shape
        seg = (y1 - y0) * nx * self.sample_bytes             # one plane's strip: contiguous in the file
        mv = memoryview(view_u8)
        fd = self._file()
        for k, z in enumerate(range(z0, z1)):
            off, got = self.img.data_offset + (z * ny + y0) * nx * self.sample_bytes, 0
            part = mv[k * seg:(k + 1) * seg]
            while got < seg:
                r = os.preadv(fd, [part[got:]], off + got)
                if r <= 0:
                    raise FITSReadError("truncated FITS payload")
                got += r
        return (z1 - z0) * seg

    def decode(self, device, stream, d_raw_ptr, nsamples, d_out_ptr):
        img = self.img
        has_blank = img.blank is not None and img.bitpix > 0
        _lib.call("spc_fits_to_f32", device, stream.handle, C.c_void_p(d_raw_ptr), img.bitpix, img.bscale, img.bzero,
                  1 if has_blank else 0, int(img.blank) if has_blank else 0, nsamples, C.c_void_p(d_out_ptr))

    def release(self):
        with self._lock:
            if self._fd is not None:
                os.close(self._fd)
                self._fd = None


# ---- the strip pipeline ------------------------------------------------------------------------------------
def plan_rows(shape, budget, mask_array=False, align=8):
    """rows per strip: two strips in flight (one computing, one being staged) + their mask strips within
    half the budget, at least `align` rows, a multiple of `align` (16-byte aligned row starts for any nx % 4 == 0)"""
    nz, ny, nx = shape
    per_row = nz * nx * (4 + (1 if mask_array else 0))
    rows = int((budget // 2) // (2 * per_row))
    rows = max(align, rows // align * align)
    return min(ny, rows)


# pinned staging buffers are expensive to make (page-locking ~0.5 GiB per pass costs as much as staging a few GiB) and
# cheap to keep: a pipeline borrows its set from here and hands it back in close()
_PINNED_IDLE = {}
_PINNED_LOCK = threading.Lock()


def _take_pinned(cap, n):
    from .io_fits import _Pinned
    with _PINNED_LOCK:
        have = _PINNED_IDLE.setdefault(cap, [])
        got = [have.pop() for _ in range(min(n, len(have)))]
    got += [_Pinned(cap) for _ in range(n - len(got))]
    for b in got:
        b.free_evt = None
    return got


def _give_pinned(cap, bufs, keep_bytes=2 << 30):
    with _PINNED_LOCK:
        have = _PINNED_IDLE.setdefault(cap, [])
        for b in bufs:
            if sum(len(v) * c for c, v in _PINNED_IDLE.items()) + cap <= keep_bytes:
                have.append(b)
            else:
                b.close()


def release_pinned():
    """unpin every idle staging buffer"""
    with _PINNED_LOCK:
        for v in _PINNED_IDLE.values():
            for b in v:
                b.close()
        _PINNED_IDLE.clear()


class StripPipeline:
    """Row strips (nz, rows, nx) of a source as device arrays, ONE continuous pipeline over the whole cube: the work is
    cut into chunks (a run of planes of one strip, <= chunk_bytes); reader threads fill pinned buffers, the iterating
    thread issues chunk i's asynchronous H2D (+ device decode) on the copy stream as soon as it is read and hands
    buffer i % nbuf to chunk i + nbuf.  The readers never drain at a strip boundary; `slots` strip buffers rotate on
    the device, and a buffer is only overwritten once the consumer's stream has passed the event recorded after ITS
    kernels (done()).  Iteration yields (y0, y1, DeviceArray) with `consumer_stream` already waiting for the strip."""

    def __init__(self, source, device, rows, consumer_stream, slots=2, chunk_bytes=None, nbuffers=None, readers=None):
        from .device import Event
        self.source, self.device, self.rows, self.consumer = source, device, int(rows), consumer_stream
        self.slots = slots
        self.chunk_bytes = int(chunk_bytes or (_env_int("SPC_STREAM_CHUNK_MB", 32) << 20))
        self.nbuf = int(nbuffers or _env_int("SPC_STREAM_BUFFERS", 16))
        self.readers = int(readers or _env_int("SPC_STREAM_READERS", 8))
        nz, ny, nx = source.shape
        self.bounds = [(y0, min(ny, y0 + self.rows)) for y0 in range(0, ny, self.rows)]
        seg = self.rows * nx * source.sample_bytes
        self.ppc = max(1, min(nz, self.chunk_bytes // max(1, seg)))      # planes per chunk
        cap = -(-(self.ppc * seg) // (1 << 20)) << 20       # whole MiB: sets of equal size are shared between passes
        self.cap = cap
        self.pinned = _take_pinned(cap, self.nbuf)
        self.d_raw = [DeviceArray((cap,), np.uint8, device) for _ in range(self.nbuf)] if source.decode else None
        self.copy = Stream(device)
        self.bufs = [DeviceArray((nz, self.rows, nx), source.out_dtype, device) for _ in range(min(slots, len(self.bounds)))]
        self.done_evt = [None] * len(self.bufs)
        self.Event = Event
        self.bytes = 0

    def close(self):
        _give_pinned(self.cap, self.pinned)
        self.pinned = []

    def __iter__(self):
        from concurrent.futures import ThreadPoolExecutor
        src, dev = self.source, self.device
        nz, ny, nx = src.shape
        tasks = [(s, z0, min(nz, z0 + self.ppc)) for s in range(len(self.bounds)) for z0 in range(0, nz, self.ppc)]
        last_of = {}
        for i, (s, _, _) in enumerate(tasks):
            last_of[s] = i
        isz = src.out_dtype.itemsize
        pool = ThreadPoolExecutor(max_workers=max(1, self.readers))
        pending = {}

        def submit(i):
            b = self.pinned[i % self.nbuf]
            if b.free_evt is not None:                       # the H2D that last used this buffer must be done
                b.free_evt.synchronize()
                b.free_evt = None
            s, z0, z1 = tasks[i]
            y0, y1 = self.bounds[s]
            pending[i] = pool.submit(src.read_into, b.view, z0, z1, y0, y1)

        try:
            nxt = 0
            while nxt < min(self.nbuf, len(tasks)):
                submit(nxt)
                nxt += 1
            for i, (s, z0, z1) in enumerate(tasks):
                n = pending.pop(i).result()
                y0, y1 = self.bounds[s]
                rows = y1 - y0
                slot = s % len(self.bufs)
                if z0 == 0 and self.done_evt[slot] is not None:      # the consumer's kernels on the strip that held this slot
                    self.copy.wait_event(self.done_evt[slot])
                    self.done_evt[slot] = None
                b = self.pinned[i % self.nbuf]
                # a short last strip is stored densely: (nz, rows, nx) at the head of the slot
                dst = self.bufs[slot].ptr + z0 * rows * nx * isz
                if src.decode is None:
                    _lib.call("spc_memcpy_h2d", dev, C.c_void_p(dst), C.c_void_p(b.ptr), C.c_size_t(n), self.copy.handle)
                else:
                    raw = self.d_raw[i % self.nbuf]
                    _lib.call("spc_memcpy_h2d", dev, C.c_void_p(raw.ptr), C.c_void_p(b.ptr), C.c_size_t(n), self.copy.handle)
                    src.decode(dev, self.copy, raw.ptr, n // src.sample_bytes, dst)
                ev = self.Event(dev)
                ev.record(self.copy)
                b.free_evt = ev
                self.bytes += n
                if nxt < len(tasks):
                    submit(nxt)
                    nxt += 1
                if i == last_of[s]:
                    self.consumer.wait_event(ev)             # the strip is complete when its last chunk has landed
                    strip = DeviceArray((nz, rows, nx), src.out_dtype, dev, ptr=self.bufs[slot].ptr, owner=self.bufs[slot])
                    yield y0, y1, strip
                    d = self.Event(dev)                      # recorded after whatever the consumer queued on its stream
                    d.record(self.consumer)
                    self.done_evt[slot] = d
        finally:
            pool.shutdown(wait=True)
            self.copy.synchronize()


def _mask_terms(cube):
    """device terms of the cube's mask, array term kept on the HOST (strips of it travel with the data)"""
    if cube._mask is None:
        return None
    terms = cube._mask._device_terms(cube)
    if terms is None:
        raise NotImplementedError("a streamed (out-of-core) cube takes masks made of isfinite / threshold comparisons on "
                                  "the cube itself and boolean arrays; this mask needs the whole cube on the host")
    flags, lo, hi, m = terms
    if m is not None:
        flags |= _lib.MASK_ARRAY
    lo = float(lo) if flags & (_lib.MASK_GT | _lib.MASK_GE) else 0.0
    hi = float(hi) if flags & (_lib.MASK_LT | _lib.MASK_LE) else 0.0
    return flags, lo, hi, m


class Strips:
    """(y0, y1, data strip, MaskSpec or None) of a streamed cube on `stream`; data and the mask's array term come
    through two pipelines in lockstep"""

    def __init__(self, cube, stream, rows=None):
        from . import ops
        self.ops = ops
        src = cube._stream_source()
        self.terms = _mask_terms(cube)
        has_arr = self.terms is not None and self.terms[3] is not None
        if rows is None:
            rows = plan_rows(src.shape, hbm_budget(cube.device), mask_array=has_arr)
        self.rows = rows
        self.data = StripPipeline(src, cube.device, rows, stream)
        self.mask = None
        if has_arr:
            m = np.broadcast_to(self.terms[3], src.shape)
            self.mask = StripPipeline(NdarraySource(m, np.uint8), cube.device, rows, stream,
                                      nbuffers=max(4, self.data.nbuf // 2), readers=max(2, self.data.readers // 2))

    @property
    def bytes(self):
        return self.data.bytes + (self.mask.bytes if self.mask is not None else 0)

    def __iter__(self):
        try:
            if self.mask is None:
                for y0, y1, dev in self.data:
                    spec = None if self.terms is None else self.ops.MaskSpec(self.terms[0], self.terms[1], self.terms[2], None)
                    yield y0, y1, dev, spec
            else:
                for (y0, y1, dev), (_, _, marr) in zip(self.data, self.mask):
                    yield y0, y1, dev, self.ops.MaskSpec(self.terms[0], self.terms[1], self.terms[2], marr)
        finally:
            self.data.close()
            if self.mask is not None:
                self.mask.close()


def _rows_view(arr, y0, y1):
    ny, nx = arr.shape
    return DeviceArray((y1 - y0, nx), arr.dtype, arr.device, ptr=arr.ptr + y0 * nx * arr.dtype.itemsize, owner=arr)


def _mask_terms(cube):
    """device terms of the cube's mask, array term kept on the HOST (strips of it travel with the data)"""
    from . import masks as M
    if cube._mask is None:
        return None
    terms = cube._mask._device_terms(cube)
    if terms is None:
        raise NotImplementedError("a streamed (out-of-core) cube takes masks made of isfinite / threshold comparisons on "
                                  "the cube itself and boolean arrays; this mask needs the whole cube on the host")
    flags, lo, hi, m = terms
    if m is not None:
        flags |= _lib.MASK_ARRAY
    lo = float(lo) if flags & (_lib.MASK_GT | _lib.MASK_GE) else 0.0
    hi = float(hi) if flags & (_lib.MASK_LT | _lib.MASK_LE) else 0.0
    return flags, lo, hi, m


_TYPES = dict(m0=np.float64, m1=np.float64, m2=np.float64, mu=np.float64, s0=np.float64, argmax=np.int64, argmin=np.int64,
              vmax=np.float32, vmin=np.float32, nvalid=np.int32)


def moments(cube, want, d_cen, dv, m1_add, kernel=None, cen_host=None, rows=None, stats=None):
    """the maps of ops.moments / ops.spectral_conv_moments for a streamed cube: {name: (ny, nx) DeviceArray}, every
    strip's kernel writing its rows of the final maps.  stats (dict) receives bytes staged and strips."""
    from . import ops
    nz, ny, nx = cube._shape
    maps = {k: DeviceArray((ny, nx), _TYPES[k], cube.device) for k in want}
    compute = Stream(cube.device)
    st = Strips(cube, compute, rows)
    need = _lib.load().spc_moments_workspace_bytes(nz, st.rows, nx)
    ws = DeviceArray((max(int(need), 1),), np.uint8, cube.device)        # ONE scratch for every strip's launch
    n = 0
    for y0, y1, dev, mspec in st:
        out = {k: _rows_view(maps[k], y0, y1) for k in want}
        if kernel is None:
            ops.moments(dev, d_cen, dv=dv, m1_add=m1_add, mask=mspec, want=want, stream=compute, out=out, workspace=ws)
        else:
            ops.spectral_conv_moments(dev, kernel, d_cen, dv=dv, m1_add=m1_add, mask=mspec, want=want, stream=compute,
                                      out=out, cen_host=cen_host)
        n += 1
    compute.synchronize()
    if stats is not None:
        stats.update(bytes=st.bytes, strips=n, rows=st.rows)
    return maps


def statistics(cube, rows=None):
    """ops.stats_global of a streamed cube: the per-strip records combined like the reference combines its chunks
    (dask_spectral_cube.py:795-814)"""
    from . import ops
    from .distributed import combine_statistics
    compute = Stream(cube.device)
    parts = []
    for y0, y1, dev, mspec in Strips(cube, compute, rows):
        parts.append(ops.stats_global(dev, mask=mspec, stream=compute))      # (waits for its own records)
    return combine_statistics(parts)
