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

What streams: moments of any order along any axis, argmax / argmin / max / min, statistics() and the
axis=None reductions, median / percentile / mad_std along the spectral axis, sigma_clip_spectrally; the cube -> cube
operators (spectral_smooth, spatial_smooth - halo strips before a moment, slabs of whole planes to a sink -,
spectral_interpolate, convolve_to, reproject) through map_strips into a host array or a FITS file, or fused with
a following moment.  What still needs the resident cube raises HugeCubeError with the budget in the message.

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


class DaskSource:
    """a dask array (nz, ny, nx) - what a DaskSpectralCube holds (dask_spectral_cube.py:85-116) - as a strip source: the
    reader threads of the strip pipeline compute the window they are asked for STRAIGHT INTO the pinned staging buffer
    (dask.array.store, synchronous scheduler inside the reader thread: the pipeline's own threads are the parallelism),
    converting to float32 on the way.  The per-chunk seam (`map_blocks` + a chunk function, dask_adapter.py) hands every
    chunk over through dask's graph machinery and a Python callable - two thirds of the wall clock at 1024^3
    (profiles/r03_dask_seam_breakdown.log); here dask only evaluates slices of the stored array."""

    def __init__(self, array, out_dtype=np.float32):
        if len(array.shape) != 3:
            raise ValueError("a (nz, ny, nx) dask array")
        self.data = array
        self.shape = tuple(int(s) for s in array.shape)
        self.out_dtype = np.dtype(out_dtype)
        self.sample_bytes = self.out_dtype.itemsize
        self.decode = None
        self.max_strip_mb = _env_int("SPC_STREAM_MAX_STRIP_MB", 1024)     # never resident, so always streamed: strips small enough to overlap
        self.preferred_chunk_mb = 64          # (measured at 1024^3, chunks (-1, 256, 256): 16 MiB windows 11 GB/s, 32: 21, 64: 25, 128: 15)

    def read_into(self, view_u8, z0, z1, y0, y1):
        import dask.array as da
        nz, ny, nx = self.shape
        n = (z1 - z0) * (y1 - y0) * nx
        dst = np.frombuffer(view_u8, dtype=self.out_dtype, count=n).reshape(z1 - z0, y1 - y0, nx)
        piece = self.data[z0:z1, y0:y1]
        if piece.dtype != self.out_dtype:
            piece = piece.astype(self.out_dtype)
        # the scheduler goes with THIS call: dask.config.set is process-global and its enter / exit is not thread-safe,
        # and read_into runs on the pipeline's reader threads
        da.store(piece, dst, lock=False, compute=True, scheduler="synchronous")
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

    def __del__(self):                   # the descriptor lives as long as the cubes that read from this source
        try:
            self.release()
        except Exception:
            pass


# ---- the strip pipeline ------------------------------------------------------------------------------------
def plan_rows(shape, budget, mask_array=False, align=8, max_strip_mb=0):
    """rows per strip: two strips in flight (one computing, one being staged) + their mask strips within
    half the budget, at least `align` rows, a multiple of `align` (16-byte aligned row starts for any nx % 4 == 0)"""
    nz, ny, nx = shape
    per_row = nz * nx * (4 + (1 if mask_array else 0))
    rows = int((budget // 2) // (2 * per_row))
    # (a source that is streamed although it would fit - a dask array: never resident - still goes in several strips, so that
    # staging, kernels and read-back of neighbouring strips overlap)
    if max_strip_mb:
        rows = min(rows, max(align, (int(max_strip_mb) << 20) // max(1, per_row)))
    rows = max(align, rows // align * align)
    return min(ny, rows)


def plan_planes(shape, budget, mask_array=False, out_factor=1.0):
    """planes per slab for the operators that need whole image planes (reprojection, statistics along y / x): two slabs
    in flight + their mask slabs + a result of `out_factor` times a slab's size within half the budget, at least one"""
    nz, ny, nx = shape
    per_plane = ny * nx * (4 + (1 if mask_array else 0) + 4.0 * out_factor)
    return int(max(1, min(nz, (budget // 2) // (2 * per_plane))))


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

    def __init__(self, source, device, rows, consumer_stream, slots=2, chunk_bytes=None, nbuffers=None, readers=None, halo=0, axis=1):
        from .device import Event
        self.source, self.device, self.rows, self.consumer = source, device, int(rows), consumer_stream
        self.halo = int(halo)          # extra rows loaded on each side of a strip (clipped at the cube's edges): spatial stencils
        self.axis = int(axis)          # 1: row strips (nz, rows, nx);  0: slabs of `rows` whole planes (rows, ny, nx)
        if self.axis not in (0, 1) or (self.axis == 0 and self.halo):
            raise ValueError("strips along y (axis 1, optional halo rows) or slabs of planes (axis 0)")
        self.slots = slots
        # (a source may ask for larger pieces: every window of a dask array costs a graph evaluation)
        pref = int(getattr(source, "preferred_chunk_mb", 0))
        self.chunk_bytes = int(chunk_bytes or (_env_int("SPC_STREAM_CHUNK_MB", pref or 32) << 20))
        self.nbuf = int(nbuffers or _env_int("SPC_STREAM_BUFFERS", 16))
        self.readers = int(readers or _env_int("SPC_STREAM_READERS", 8))
        nz, ny, nx = source.shape
        n_ax = ny if self.axis == 1 else nz
        self.bounds = [(a, min(n_ax, a + self.rows)) for a in range(0, n_ax, self.rows)]
        if self.axis == 1:
            self.loaded = [(max(0, y0 - self.halo), min(ny, y1 + self.halo)) for y0, y1 in self.bounds]     # rows a strip holds
            self.ext = min(ny, self.rows + 2 * self.halo)
            seg = self.ext * nx * source.sample_bytes
            self.ppc = max(1, min(nz, self.chunk_bytes // max(1, seg)))      # planes per chunk
            shape = (nz, self.ext, nx)
        else:
            self.loaded = [(0, ny)] * len(self.bounds)
            self.ext = ny
            seg = ny * nx * source.sample_bytes
            self.ppc = max(1, min(self.rows, nz, self.chunk_bytes // max(1, seg)))
            shape = (min(self.rows, nz), ny, nx)
        cap = -(-(self.ppc * seg) // (1 << 20)) << 20       # whole MiB: sets of equal size are shared between passes
        self.cap = cap
        self.pinned = _take_pinned(cap, self.nbuf)
        self.d_raw = [DeviceArray((cap,), np.uint8, device) for _ in range(self.nbuf)] if source.decode else None
        self.copy = Stream(device)
        self.bufs = [DeviceArray(shape, source.out_dtype, device) for _ in range(min(slots, len(self.bounds)))]
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
        if self.axis == 1:
            tasks = [(s, z0, min(nz, z0 + self.ppc)) for s in range(len(self.bounds)) for z0 in range(0, nz, self.ppc)]
        else:
            tasks = [(s, z0, min(b, z0 + self.ppc)) for s, (a, b) in enumerate(self.bounds) for z0 in range(a, b, self.ppc)]
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
            y0, y1 = self.loaded[s]
            pending[i] = pool.submit(src.read_into, b.view, z0, z1, y0, y1)

        try:
            nxt = 0
            while nxt < min(self.nbuf, len(tasks)):
                submit(nxt)
                nxt += 1
            for i, (s, z0, z1) in enumerate(tasks):
                n = pending.pop(i).result()
                y0, y1 = self.bounds[s]
                h0, h1 = self.loaded[s]
                rows = h1 - h0
                slot = s % len(self.bufs)
                first_z = 0 if self.axis == 1 else y0
                if z0 == first_z and self.done_evt[slot] is not None:      # the consumer's kernels on the strip that held this slot
                    self.copy.wait_event(self.done_evt[slot])
                    self.done_evt[slot] = None
                b = self.pinned[i % self.nbuf]
                # a short last strip is stored densely: (nz, rows, nx) at the head of the slot
                dst = self.bufs[slot].ptr + (z0 - first_z) * rows * nx * isz
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
                    strip = DeviceArray((nz, rows, nx) if self.axis == 1 else (y1 - y0, ny, nx), src.out_dtype, dev,
                                        ptr=self.bufs[slot].ptr, owner=self.bufs[slot])
                    strip.top = (y0 - h0) if self.axis == 1 else 0     # the strip's own rows are [top, top + (y1 - y0)) of what was loaded
                    strip.z0 = y0 if self.axis == 0 else 0            # (a slab's first channel)
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
    cached = getattr(cube, "_stream_terms", None)
    if cached is not None and cached[0] is cube._mask:        # (an inverted / composite boolean mask materialises a host
        return cached[1]                                      # array: once per cube, not once per caller)
    terms = cube._mask._device_terms(cube)
    if terms is None:
        raise NotImplementedError("a streamed (out-of-core) cube takes masks made of isfinite / threshold comparisons on "
                                  "the cube itself and boolean arrays; this mask needs the whole cube on the host")
    flags, lo, hi, m = terms
    if m is not None:
        flags |= _lib.MASK_ARRAY
    lo = float(lo) if flags & (_lib.MASK_GT | _lib.MASK_GE) else 0.0
    hi = float(hi) if flags & (_lib.MASK_LT | _lib.MASK_LE) else 0.0
    try:
        cube._stream_terms = (cube._mask, (flags, lo, hi, m))
    except AttributeError:
        pass
    return flags, lo, hi, m


class Strips:
    """(y0, y1, data strip, MaskSpec or None) of a streamed cube on `stream`; data and the mask's array term come
    through two pipelines in lockstep"""

    def __init__(self, cube, stream, rows=None, halo=0, axis=1, out_factor=1.0):
        from . import ops
        self.ops = ops
        src = cube._stream_source()
        self.terms = _mask_terms(cube)
        has_arr = self.terms is not None and self.terms[3] is not None
        if rows is None:
            rows = (plan_rows(src.shape, hbm_budget(cube.device), mask_array=has_arr, max_strip_mb=getattr(src, "max_strip_mb", 0)) if axis == 1 else
                    plan_planes(src.shape, hbm_budget(cube.device), mask_array=has_arr, out_factor=out_factor))
        self.rows = rows
        self.data = StripPipeline(src, cube.device, rows, stream, halo=halo, axis=axis)
        self.mask = None
        if has_arr:
            m = np.broadcast_to(self.terms[3], src.shape)
            self.mask = StripPipeline(NdarraySource(m, np.uint8), cube.device, rows, stream,
                                      nbuffers=max(4, self.data.nbuf // 2), readers=max(2, self.data.readers // 2), halo=halo, axis=axis)

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


def original_include(cube, dev, mspec, stream):
    """the mask of the streamed cube *cube* evaluated on ITS strip `dev`, as an array-only MaskSpec: what a derived cube
    that keeps its parent's mask (spectral_smooth / spatial_smooth results, cube.py `_mask_spec`) is reduced and filled
    with - the isfinite / threshold terms are bound to the parent's voxels, not to the operator's output"""
    from . import masks as M, ops
    if mspec is None:
        return None
    nan_excluded = M.contains(cube._mask, M.NotNaNMask)
    if mspec.array is not None and not nan_excluded and not (mspec.flags & ~_lib.MASK_ARRAY):
        return mspec
    inc = ops.mask_include(dev, mspec, nan_excluded=nan_excluded, stream=stream)
    return ops.MaskSpec(_lib.MASK_ARRAY, 0.0, 0.0, inc)


_TYPES = dict(m0=np.float64, m1=np.float64, m2=np.float64, mu=np.float64, s0=np.float64, argmax=np.int64, argmin=np.int64,
              vmax=np.float32, vmin=np.float32, nvalid=np.int32)


def moments(cube, want, d_cen, dv, m1_add, kernel=None, cen_host=None, rows=None, stats=None, pre=None, halo=0):
    """the maps of ops.moments / ops.spectral_conv_moments for a streamed cube: {name: (ny, nx) DeviceArray}, every
    strip's kernel writing its rows of the final maps.  stats (dict) receives bytes staged and strips.
    pre(strip, mask spec, stream) -> strip of the same shape: an operator applied to the strip before it is reduced (the
    smoothed cube of spatial_smooth(...).moment(...): strips carry `halo` extra rows per side, which are smoothed against
    an artificial edge and never reduced; the smoothed cube keeps the ORIGINAL mask)."""
    from . import ops
    nz, ny, nx = cube._shape
    maps = {k: DeviceArray((ny, nx), _TYPES[k], cube.device) for k in want}
    compute = Stream(cube.device)
    if rows is None and pre is not None:
        terms = _mask_terms(cube)
        per_row = nz * nx * (4 + (1 if terms is not None and terms[3] is not None else 0) + 4)      # + the operator's result
        rows = max(8, int((hbm_budget(cube.device) // 2) // (2 * per_row)) // 8 * 8 - 2 * halo)
        rows = max(8, min(ny, rows))
    st = Strips(cube, compute, rows, halo=halo)
    need = _lib.load().spc_moments_workspace_bytes(nz, st.rows, nx)
    ws = DeviceArray((max(int(need), 1),), np.uint8, cube.device)        # ONE scratch for every strip's launch
    n = 0
    for y0, y1, dev, mspec in st:
        out = {k: _rows_view(maps[k], y0, y1) for k in want}
        if pre is not None:
            top = getattr(dev, "top", 0)
            keep = original_include(cube, dev, mspec, compute)      # (evaluated on the PARENT's voxels, before `pre`)
            sm = pre(dev, mspec, compute)
            ops.moments(sm.rows(top, top + (y1 - y0)), d_cen, dv=dv, m1_add=m1_add,
                        mask=keep.rows(top, top + (y1 - y0)) if keep is not None else None,
                        want=want, stream=compute, out=out, workspace=ws)
            compute.synchronize()          # `sm` goes back to the pool when the next strip replaces it
        elif kernel is None:
            ops.moments(dev, d_cen, dv=dv, m1_add=m1_add, mask=mspec, want=want, stream=compute, out=out, workspace=ws)
        else:
            ops.spectral_conv_moments(dev, kernel, d_cen, dv=dv, m1_add=m1_add, mask=mspec, want=want, stream=compute,
                                      out=out, cen_host=cen_host)
        n += 1
    compute.synchronize()
    if stats is not None:
        stats.update(bytes=st.bytes, strips=n, rows=st.rows)
    return maps


def percentile_axis0(cube, q, center=None, scale=1.0, rows=None):
    """ops.percentile_axis0 of a streamed cube (median / percentile / the two selections of mad_std along the spectral
    axis: every spaxel is whole in a row strip): (ny, nx) float32 DeviceArray.  center: the (ny, nx) float32 map of the
    first selection (mad_std), read row strip by row strip"""
    from . import ops
    nz, ny, nx = cube._shape
    out = DeviceArray((ny, nx), np.float32, cube.device)
    compute = Stream(cube.device)
    for y0, y1, dev, mspec in Strips(cube, compute, rows):
        ops.percentile_axis0(dev, q, mask=mspec, center=_rows_view(center, y0, y1) if center is not None else None, scale=scale,
                             stream=compute, out=_rows_view(out, y0, y1))
    compute.synchronize()
    return out


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


# ---- cube -> cube operators of a streamed cube: result strips back to the host ---------------------------------
class NdarraySink:
    """(nz, ny, nx) float32 host array / memory map receiving row strips"""

    swap = False

    def __init__(self, array):
        if array.dtype != np.float32 or array.ndim != 3:
            raise TypeError("the result array must be float32 (nz, ny, nx)")
        self.array = array
        self.shape = tuple(array.shape)

    def write(self, view_u8, z0, z1, y0, y1):
        nx = self.shape[2]
        n = (z1 - z0) * (y1 - y0) * nx
        np.copyto(self.array[z0:z1, y0:y1], np.frombuffer(view_u8, dtype=np.float32, count=n).reshape(z1 - z0, y1 - y0, nx))

    def close(self):
        if hasattr(self.array, "flush"):
            self.array.flush()


class FitsSink:
    """BITPIX = -32 FITS file of a given shape, written strip by strip: the header and the (sparse) payload are laid down
    first, every chunk arrives byte-swapped by the device and goes out with one os.pwrite per plane segment (the strip's
    rows of a plane are contiguous in the file)"""

    swap = True

    def __init__(self, path, header, shape, overwrite=False):
        from . import io_fits
        if os.path.exists(path) and not overwrite:
            raise OSError("File %r already exists (use overwrite=True)" % path)
        self.shape = tuple(int(s) for s in shape)
        nz, ny, nx = self.shape
        hdr = io_fits.parse_header(header) if header is not None else {}
        cards = [io_fits._card("SIMPLE", True), io_fits._card("BITPIX", -32), io_fits._card("NAXIS", 3), io_fits._card("NAXIS1", nx),
                 io_fits._card("NAXIS2", ny), io_fits._card("NAXIS3", nz)]
        for k, v in hdr.items():
            if k in ("SIMPLE", "BITPIX", "BSCALE", "BZERO", "BLANK", "EXTEND", "WCSAXES") or k.startswith("NAXIS"):
                continue
            cards.append(io_fits._card(k, v))
        cards.append("END".ljust(80))
        text = "".join(cards)
        text += " " * ((-len(text)) % io_fits.BLOCK)
        self.base = len(text)
        total = nz * ny * nx * 4
        # The payload goes to a sibling file that takes the target's name when the last strip has landed: the target may
        # be the very file the strips are read from (cube.write(path, overwrite=True) of a streamed cube read from path:
        # O_TRUNC on it would destroy the input before the first strip is read), and a failed run leaves no half-written
        # file under the target's name.
        # A symlinked target is written THROUGH the link (the rename lands on the file the link names), an existing
        # target keeps its permission bits, and part files a killed run of this process id left behind are swept.
        self.path = os.path.realpath(os.fspath(path))
        mode = 0o644
        try:
            mode = os.stat(self.path).st_mode & 0o7777
        except OSError:
            pass
        self._sweep_stale_parts()
        self.part = "%s.spc-part-%d-%x" % (self.path, os.getpid(), id(self) & 0xffffff)
        self.fd = os.open(self.part, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
        try:
            os.fchmod(self.fd, mode)
        except OSError:
            pass
        size = self.base + total + (-total) % io_fits.BLOCK                       # zero padding included
        try:
            os.pwrite(self.fd, text.encode("ascii"), 0)
            try:
                # blocks reserved up front: the writer threads then fill allocated pages instead of growing a sparse file
                # one page fault at a time (tmpfs / ext4: the extents are laid out once, contiguously)
                os.posix_fallocate(self.fd, 0, size)
            except OSError:
                os.ftruncate(self.fd, size)
            # The strips arrive as one piece per plane (rows x nx samples): with the 64-row strips of a 2 GiB test budget that is
            # 256 KiB per os.pwrite, 32768 calls for 8 GiB, each under the file's write lock: 4.7 GB/s where 64 MiB writes reach
            # 15 - 20 GB/s on the same file system (round-3 scratch scripts, in the history: tools/README.md).  A shared mapping
            # (SPC_FITS_SINK_MMAP=1) lets the writer threads copy side by side, but a fresh file then costs a page-cache fault
            # per 4 KiB: 4.1 GB/s - no better, so the pwrite form stays the default.  The pieces grow with the strip: a cube
            # that is out of core for 288 GB of HBM has strips of hundreds of rows (10+ MiB per piece).
            self.map = None
            if os.environ.get("SPC_FITS_SINK_MMAP", "0") == "1":
                import mmap
                self.mm = mmap.mmap(self.fd, size, access=mmap.ACCESS_WRITE)
                self.map = np.frombuffer(self.mm, dtype=np.uint8)
        except BaseException:
            self.abort()
            raise

    def _sweep_stale_parts(self):
        """part files beside the target whose writer is gone (the pid in the name is not alive): a hard kill leaves them"""
        d, base = os.path.dirname(self.path) or ".", os.path.basename(self.path) + ".spc-part-"
        try:
            names = [n for n in os.listdir(d) if n.startswith(base)]
        except OSError:
            return
        for n in names:
            try:
                pid = int(n[len(base):].split("-")[0])
            except ValueError:
                continue
            if pid == os.getpid():
                continue
            try:
                os.kill(pid, 0)                      # signal 0: existence check only
                continue                             # alive (or not ours to judge): leave it
            except ProcessLookupError:
                pass
            except OSError:
                continue
            try:
                os.unlink(os.path.join(d, n))
            except OSError:
                pass

    def write(self, view_u8, z0, z1, y0, y1):
        nz, ny, nx = self.shape
        seg = (y1 - y0) * nx * 4
        if self.map is not None:
            src = np.frombuffer(view_u8, dtype=np.uint8, count=(z1 - z0) * seg)
            for k, z in enumerate(range(z0, z1)):
                off = self.base + (z * ny + y0) * nx * 4
                np.copyto(self.map[off:off + seg], src[k * seg:(k + 1) * seg])
            return
        mv = memoryview(view_u8)
        for k, z in enumerate(range(z0, z1)):
            part, done, off = mv[k * seg:(k + 1) * seg], 0, self.base + (z * ny + y0) * nx * 4
            while done < seg:
                done += os.pwrite(self.fd, part[done:], off + done)

    def _unmap(self):
        if getattr(self, "map", None) is not None:
            self.map = None
            try:
                self.mm.close()
            except (BufferError, ValueError):
                pass
            self.mm = None

    def close(self, ok=True):
        if self.fd is not None:
            self._unmap()
            os.close(self.fd)
            self.fd = None
            if ok:
                os.replace(self.part, self.path)
            else:
                self.abort()

    def abort(self):
        if self.fd is not None:
            self._unmap()
            os.close(self.fd)
            self.fd = None
        try:
            os.unlink(self.part)
        except OSError:
            pass


class StripWriter:
    """result strips (nz, rows, nx) float32 from the device into a sink: chunks of planes come down through pinned
    buffers on their own stream (after the event the producer recorded), writer threads hand them to the sink while the
    next strip is being staged and computed"""

    def __init__(self, sink, device, nbuffers=8, chunk_bytes=32 << 20, writers=8):
        from concurrent.futures import ThreadPoolExecutor
        from .device import Event
        self.sink, self.device, self.Event = sink, device, Event
        self.chunk_bytes, self.nbuf = int(chunk_bytes), int(nbuffers)
        self.cap = None
        self.pinned, self.busy = [], []
        self.d_swap = []
        self.down = Stream(device)
        self.pool = ThreadPoolExecutor(max_workers=writers)
        self.i = 0
        self.keep = []                   # (event, strip) pairs: a strip stays alive until its last chunk has left the device
        self.bytes = 0

    def put(self, y0, y1, strip, produced_on, z_base=0):
        """queue the strip (rows [y0, y1) of the result's planes z_base ..); `produced_on`: the stream its kernel ran on"""
        nz, rows, nx = strip.shape
        seg = rows * nx * 4
        ppc = max(1, min(nz, self.chunk_bytes // seg))
        if self.cap is None or ppc * seg > self.cap:
            if self.pinned:
                self._drain()
                _give_pinned(self.cap, self.pinned)
            self.cap = -(-(ppc * seg) // (1 << 20)) << 20
            self.pinned = _take_pinned(self.cap, self.nbuf)
            self.busy = [None] * self.nbuf
            self.d_swap = [DeviceArray((self.cap,), np.uint8, self.device) for _ in range(self.nbuf)] if self.sink.swap else []
        ev = self.Event(self.device)
        ev.record(produced_on)
        self.down.wait_event(ev)
        last = None
        for z0 in range(0, nz, ppc):
            z1 = min(nz, z0 + ppc)
            k = self.i % self.nbuf
            self.i += 1
            if self.busy[k] is not None:
                self.busy[k].result()            # the writer that last used this buffer (re-raises its error)
            b, n = self.pinned[k], (z1 - z0) * seg
            src = strip.ptr + z0 * seg
            if self.sink.swap:                    # BITPIX -32 byte swap on the device (its own inverse)
                _lib.call("spc_fits_to_f32", self.device, self.down.handle, C.c_void_p(src), -32, 1.0, 0.0, 0, 0, n // 4,
                          C.c_void_p(self.d_swap[k].ptr))
                src = self.d_swap[k].ptr
            _lib.call("spc_memcpy_d2h", self.device, C.c_void_p(b.ptr), C.c_void_p(src), C.c_size_t(n), self.down.handle)
            done = self.Event(self.device)
            done.record(self.down)
            last = done
            self.busy[k] = self.pool.submit(self._write, done, b, z_base + z0, z_base + z1, y0, y1)
            self.bytes += n
        self.keep.append((last, strip))
        while len(self.keep) > 2:                 # at most two result strips wait on the device
            evk, _ = self.keep.pop(0)
            evk.synchronize()

    def _write(self, done, b, z0, z1, y0, y1):
        done.synchronize()
        self.sink.write(b.view, z0, z1, y0, y1)

    def _drain(self):
        for f in self.busy:
            if f is not None:
                f.result()
        self.busy = [None] * len(self.busy)

    def close(self, ok=True):
        """ok=False: the producer failed - a sink that knows how (FitsSink) discards what was written"""
        try:
            self._drain()
            self.down.synchronize()
        except BaseException:
            ok = False
            raise
        finally:
            self.pool.shutdown(wait=True)
            if self.pinned:
                _give_pinned(self.cap, self.pinned)
                self.pinned = []
            self.keep = []
            if ok or not hasattr(self.sink, "abort"):
                self.sink.close()
            else:
                self.sink.abort()


def map_strips(cube, fn, nz_out, sink, rows=None, stats=None, halo=0):
    """out[:, y0:y1] = fn(strip, mask spec, stream) for every row strip of a streamed cube; fn returns a float32
    (nz_out, rows, nx) DeviceArray produced on `stream`.  The operators this serves work per spaxel (spectral_smooth,
    spectral_interpolate, sigma_clip_spectrally, the plain filled copy): no halo."""
    nz, ny, nx = cube._shape
    if tuple(sink.shape) != (nz_out, ny, nx):
        raise ValueError("the sink has shape %s, the result %s" % (tuple(sink.shape), (nz_out, ny, nx)))
    compute = Stream(cube.device)
    if rows is None:
        src = cube._stream_source()
        terms = _mask_terms(cube)
        # two input strips + two result strips (+ the operator's own scratch) within half the budget
        per_row = nx * (nz * (4 + (1 if terms is not None and terms[3] is not None else 0)) + 2 * nz_out * 4)
        rows = max(8, int((hbm_budget(cube.device) // 2) // (2 * per_row)) // 8 * 8 - 2 * halo)
        if getattr(src, "max_strip_mb", 0):
            rows = min(rows, max(8, ((int(src.max_strip_mb) << 20) // max(1, nx * nz * 4)) // 8 * 8))
        rows = max(8, min(src.shape[1], rows))
    st = Strips(cube, compute, rows, halo=halo)
    w = StripWriter(sink, cube.device)
    n, done = 0, False
    try:
        for y0, y1, dev, mspec in st:
            res = fn(dev, mspec, compute)
            top = getattr(dev, "top", 0)
            if halo and res.shape[1] != y1 - y0:
                # the strip's own rows of the extended result, compacted on the device (the halo rows saw an artificial edge)
                own = DeviceArray((res.shape[0], y1 - y0, nx), np.float32, cube.device)
                row = nx * 4
                _lib.call("spc_memcpy3d_d2d", cube.device, C.c_void_p(own.ptr), row, (y1 - y0) * row,
                          C.c_void_p(res.ptr + top * row), row, res.shape[1] * row, row, y1 - y0, res.shape[0], compute.handle)
                res = own
            w.put(y0, y1, res, compute)
            n += 1
        done = True
    finally:
        w.close(ok=done)
    if stats is not None:
        stats.update(bytes_in=st.bytes, bytes_out=w.bytes, strips=n, rows=st.rows)


def map_slabs(cube, fn, out_yx, sink, planes=None, stats=None):
    """out[z0:z1] = fn(slab, mask spec, stream) for every slab of whole planes of a streamed cube; fn returns a float32
    (z1 - z0, ny_out, nx_out) DeviceArray produced on `stream`.  The operators this serves work per channel
    (reprojection of the celestial plane): the channels of a slab are independent."""
    nz, ny, nx = cube._shape
    ny_out, nx_out = out_yx
    if tuple(sink.shape) != (nz, ny_out, nx_out):
        raise ValueError("the sink has shape %s, the result %s" % (tuple(sink.shape), (nz, ny_out, nx_out)))
    compute = Stream(cube.device)
    st = Strips(cube, compute, planes, axis=0, out_factor=2.0 * ny_out * nx_out / float(ny * nx))
    w = StripWriter(sink, cube.device)
    n, done = 0, False
    try:
        for z0, z1, dev, mspec in st:
            w.put(0, ny_out, fn(dev, mspec, compute), compute, z_base=z0)
            n += 1
        done = True
    finally:
        w.close(ok=done)
    if stats is not None:
        stats.update(bytes_in=st.bytes, bytes_out=w.bytes, slabs=n, planes=st.rows)


def slab_maps(cube, fn, names, width, dtypes, planes=None):
    """{name: (nz, width) DeviceArray}: maps whose FIRST axis is the spectral one (statistics along y or x of a streamed
    cube), assembled on the device slab by slab: fn(slab, mask spec, stream, {name: rows z0:z1 of the map}, z0, z1)."""
    nz = cube._shape[0]
    maps = {k: DeviceArray((nz, width), dtypes[k], cube.device) for k in names}
    compute = Stream(cube.device)
    for z0, z1, dev, mspec in Strips(cube, compute, planes, axis=0, out_factor=0.0):
        fn(dev, mspec, compute, {k: _rows_view(maps[k], z0, z1) for k in names}, z0, z1)
    compute.synchronize()
    return maps


def stats_axis(cube, axis, want):
    """ops.stats_axis of a streamed cube: along the spectral axis strip by strip (every spaxel whole), along y or x
    slab by slab (every image plane whole); the maps are assembled on the device"""
    from . import ops
    nz, ny, nx = cube._shape
    dt = ops._STAT_DTYPES
    if axis == 0:
        maps = {k: DeviceArray((ny, nx), dt[k], cube.device) for k in want}
        compute = Stream(cube.device)
        for y0, y1, dev, mspec in Strips(cube, compute):
            ops.stats_axis(dev, 0, mask=mspec, want=want, stream=compute, out={k: _rows_view(maps[k], y0, y1) for k in want})
        compute.synchronize()
        return maps
    return slab_maps(cube, lambda dev, mspec, stream, out, z0, z1: ops.stats_axis(dev, axis, mask=mspec, want=want, stream=stream, out=out),
                     want, nx if axis == 1 else ny, dt)


def stats_planes(cube):
    """ops.stats_planes (one record per channel) of a streamed cube, slab by slab"""
    from . import ops
    compute = Stream(cube.device)
    parts = [ops.stats_planes(dev, mask=mspec, stream=compute) for z0, z1, dev, mspec in Strips(cube, compute, axis=0, out_factor=0.0)]
    return {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
