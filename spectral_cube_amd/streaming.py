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


# ---- sources -----------------------------------------------------------------------------------------
class NdarraySource:
    """host array / numpy memmap, (nz, ny, nx), any real dtype"""

    def __init__(self, data):
        self.data = data
        self.shape = tuple(int(s) for s in data.shape)

    def load_rows(self, y0, y1, device, stream):
        nz, ny, nx = self.shape
        rows = y1 - y0
        out = DeviceArray((nz, rows, nx), np.float32, device)
        a = self.data
        if a.dtype == np.float32 and a.flags.c_contiguous:
            # strided H2D straight from the array: rows y0:y1 of every plane
            base = a.ctypes.data + y0 * nx * 4
            _lib.call("spc_memcpy3d_h2d", device, C.c_void_p(out.ptr), nx * 4, rows * nx * 4, C.c_void_p(base),
                      nx * 4, ny * nx * 4, nx * 4, rows, nz, stream.handle)
            stream.synchronize()
        else:
            step = max(1, (64 << 20) // max(1, rows * nx * 4))
            for z0 in range(0, nz, step):
                z1 = min(nz, z0 + step)
                blk = np.ascontiguousarray(a[z0:z1, y0:y1], dtype=np.float32)
                _lib.call("spc_memcpy_h2d", device, C.c_void_p(out.ptr + z0 * rows * nx * 4), blk.ctypes.data_as(C.c_void_p),
                          blk.nbytes, stream.handle)
                stream.synchronize()
        return out


class FitsSource:
    """image HDU of a FITS file; strips come through io_fits.load_cube(rows=...) with ONE set of pinned staging
    buffers kept for the whole pass"""

    def __init__(self, path, hdu=None):
        from . import io_fits
        self.path, self.hdu = os.fspath(path), hdu
        self.img = io_fits.find_image(self.path, hdu)
        self.shape = tuple(io_fits.cube_shape(self.img))
        self._staging = {}

    def load_rows(self, y0, y1, device, stream):
        from . import io_fits
        st = self._staging.get(device)
        if st is None:
            st = self._staging[device] = io_fits.Staging(device)
        dev, _ = io_fits.load_cube(self.path, device=device, hdu=self.hdu, rows=(y0, y1), staging=st)
        return dev

    def release(self):
        for st in self._staging.values():
            st.close()
        self._staging = {}


# ---- the strip loop ----------------------------------------------------------------------------------
def plan_rows(shape, budget, mask_array=False, align=8):
    """rows per strip: two strips in flight (one computing, one being staged) + their mask strips within
    half the budget, at least `align` rows, a multiple of `align` (16-byte aligned row starts for any nx % 4 == 0)"""
    nz, ny, nx = shape
    per_row = nz * nx * (4 + (1 if mask_array else 0))
    rows = int((budget // 2) // (2 * per_row))
    rows = max(align, rows // align * align)
    return min(ny, rows)


class Strips:
    """iterate over (y0, y1, DeviceArray strip, MaskSpec strip or None); strip k + 1 is staged by a worker thread
    while the caller works on strip k."""

    def __init__(self, source, device, rows, mask_terms=None):
        self.source, self.device, self.rows = source, device, int(rows)
        self.mask_terms = mask_terms          # (flags, lo, hi, host bool array or None)
        self.copy_stream = Stream(device)
        self.bytes = 0

    def _mask_strip(self, y0, y1):
        from . import ops
        if self.mask_terms is None:
            return None
        flags, lo, hi, m = self.mask_terms
        arr = None
        if m is not None:
            nz, ny, nx = self.source.shape
            host = np.ascontiguousarray(np.broadcast_to(m, (nz, ny, nx))[:, y0:y1]).view(np.uint8)
            arr = DeviceArray.from_numpy(host, self.device, self.copy_stream)
            self.bytes += host.nbytes
        return ops.MaskSpec(flags, lo, hi, arr)

    def _stage(self, y0, y1, box):
        try:
            dev = self.source.load_rows(y0, y1, self.device, self.copy_stream)
            self.bytes += dev.nbytes
            box.append((dev, self._mask_strip(y0, y1)))
        except BaseException as exc:          # handed to the consumer
            box.append(exc)

    def __iter__(self):
        ny = self.source.shape[1]
        bounds = [(y0, min(ny, y0 + self.rows)) for y0 in range(0, ny, self.rows)]
        box, th = [], None

        def start(b):
            nonlocal box, th
            box = []
            th = threading.Thread(target=self._stage, args=(b[0], b[1], box), daemon=True)
            th.start()

        start(bounds[0])
        for i, (y0, y1) in enumerate(bounds):
            th.join()
            got = box[0]
            if isinstance(got, BaseException):
                raise got
            if i + 1 < len(bounds):
                start(bounds[i + 1])          # H2D of the next strip under the kernels of this one
            yield y0, y1, got[0], got[1]


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


def strips_of(cube, rows=None):
    src = cube._stream_source()
    terms = _mask_terms(cube)
    if rows is None:
        rows = plan_rows(src.shape, hbm_budget(cube.device), mask_array=terms is not None and terms[3] is not None)
    return Strips(src, cube.device, rows, terms)


_TYPES = dict(m0=np.float64, m1=np.float64, m2=np.float64, mu=np.float64, s0=np.float64, argmax=np.int64, argmin=np.int64,
              vmax=np.float32, vmin=np.float32, nvalid=np.int32)


def moments(cube, want, d_cen, dv, m1_add, kernel=None, cen_host=None, rows=None, stats=None):
    """the maps of ops.moments / ops.spectral_conv_moments for a streamed cube: {name: (ny, nx) DeviceArray}, every
    strip's kernel writing its rows of the final maps.  stats (dict) receives bytes staged and strips."""
    from . import ops
    nz, ny, nx = cube._shape
    maps = {k: DeviceArray((ny, nx), _TYPES[k], cube.device) for k in want}
    compute = Stream(cube.device)
    st = strips_of(cube, rows)
    n = 0
    keep = None
    for y0, y1, dev, mspec in st:
        out = {k: _rows_view(maps[k], y0, y1) for k in want}
        if kernel is None:
            ops.moments(dev, d_cen, dv=dv, m1_add=m1_add, mask=mspec, want=want, stream=compute, out=out)
        else:
            ops.spectral_conv_moments(dev, kernel, d_cen, dv=dv, m1_add=m1_add, mask=mspec, want=want, stream=compute,
                                      out=out, cen_host=cen_host)
        compute.synchronize()        # the strip's buffers go back to the pool only when its kernels are done
        keep = (dev, mspec)
        n += 1
    del keep
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
    for y0, y1, dev, mspec in strips_of(cube, rows):
        parts.append(ops.stats_global(dev, mask=mspec, stream=compute))
    return combine_statistics(parts)
