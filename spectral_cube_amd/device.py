"""HBM buffers, streams and events on top of the C ABI (no torch, no numpy
device arrays - plain pointers owned by small RAII objects)."""
import ctypes as C

import numpy as np

from . import _lib


class Stream:
    def __init__(self, device=0):
        self.device = device
        h = C.c_void_p()
        _lib.call("spc_stream_create", device, C.byref(h))
        self.handle = h

    def synchronize(self):
        _lib.call("spc_stream_sync", self.device, self.handle)

    def wait_event(self, event):
        """make later work on this stream wait for *event* (hipStreamWaitEvent)"""
        _lib.call("spc_stream_wait_event", self.device, self.handle, event.handle)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.call("spc_stream_destroy", self.device, self.handle)
                self.handle = None
        except Exception:
            pass


def _sh(stream):
    return stream.handle if isinstance(stream, Stream) else stream


class _PinnedPool:
    """Page-locked host buffers for results that come back from the device (DeviceArray.get()).  A device-to-host
    copy into fresh pageable numpy memory runs at ~3 GB/s (page faults + the runtime's staging) - 10 ms for a
    2048 x 2048 float64 moment map, more than the kernel that produced it; into pinned memory it runs at the link
    rate.  get() therefore hands out numpy arrays that live IN pinned buffers; a buffer returns to this pool when the
    last array viewing it is collected, and is unpinned once the pool holds more than *max_bytes* idle.
    Page-locked memory cannot be swapped: the bytes OUTSTANDING (handed out and still referenced - a caller may keep
    lists of maps or dask chunk outputs) are capped as well (*max_outstanding*, SPC_PINNED_MAX_BYTES, default 4 GiB);
    beyond the cap array() raises MemoryError and get() falls back to pageable numpy memory."""

    MIN_BYTES = 1 << 20          # smaller results use plain numpy memory
    MAX_BYTES = 256 << 20        # larger ones come down in chunks through the staging buffers (DeviceArray._get_chunked):
                                 # page-locking a fresh 1 GiB buffer costs ~150 ms, more than the copy it would speed up

    def __init__(self, max_bytes=1 << 30, max_outstanding=None):
        import os
        import threading
        self.max_bytes = max_bytes
        self.max_outstanding = int(os.environ.get("SPC_PINNED_MAX_BYTES", 4 << 30)) if max_outstanding is None else max_outstanding
        self.idle = {}           # size class -> [ptr]
        self.idle_bytes = 0
        self.outstanding = 0
        self.lock = threading.Lock()
        self._ctype = {}         # one ctypes array type per SIZE CLASS (a type per distinct size would be cached for ever)

    @staticmethod
    def size_class(nbytes):
        return 1 << max(int(nbytes) - 1, 1).bit_length()

    def take(self, nbytes):
        cls = self.size_class(nbytes)
        with self.lock:
            if self.outstanding + cls > self.max_outstanding:
                raise MemoryError("pinned result buffers outstanding: %d bytes (cap %d)" % (self.outstanding, self.max_outstanding))
            self.outstanding += cls
            lst = self.idle.get(cls)
            if lst:
                self.idle_bytes -= cls
                return lst.pop(), cls
        p = C.c_void_p()
        try:
            _lib.call("spc_host_alloc", C.c_size_t(cls), C.byref(p))
        except Exception:
            with self.lock:
                self.outstanding -= cls
            raise
        return p.value, cls

    def give(self, ptr, cls):
        try:
            with self.lock:
                self.outstanding -= cls
                if self.idle_bytes + cls <= self.max_bytes:
                    self.idle.setdefault(cls, []).append(ptr)
                    self.idle_bytes += cls
                    return
            _lib.call("spc_host_free", C.c_void_p(ptr))
        except Exception:        # interpreter shutdown
            pass

    def array(self, shape, dtype):
        """uninitialised numpy array of *shape* / *dtype* in a pinned buffer of the pool"""
        import weakref
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        ptr, cls = self.take(nbytes)
        ctype = self._ctype.get(cls)
        if ctype is None:
            ctype = self._ctype[cls] = C.c_byte * cls
        raw = ctype.from_address(ptr)
        weakref.finalize(raw, self.give, ptr, cls)       # numpy keeps `raw` alive as the base of every view
        return np.frombuffer(raw, dtype=dtype, count=nbytes // dtype.itemsize).reshape(shape)


pinned_pool = _PinnedPool()


class Event:
    def __init__(self, device=0):
        self.device = device
        h = C.c_void_p()
        _lib.call("spc_event_create", device, C.byref(h))
        self.handle = h

    def record(self, stream=None):
        _lib.call("spc_event_record", self.device, self.handle, _sh(stream))

    def synchronize(self):
        _lib.call("spc_event_sync", self.device, self.handle)

    def elapsed_ms(self, later):
        ms = C.c_float(0)
        _lib.call("spc_event_elapsed_ms", self.device, self.handle, later.handle, C.byref(ms))
        return ms.value

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.call("spc_event_destroy", self.device, self.handle)
                self.handle = None
        except Exception:
            pass


def synchronize(device=0):
    _lib.call("spc_device_sync", device)


def pool_stats(device=0):
    """(live_bytes, idle_bytes) of the library's device buffer pool (spc_pool_stats)."""
    live, idle = C.c_int64(0), C.c_int64(0)
    _lib.call("spc_pool_stats", device, C.byref(live), C.byref(idle))
    return live.value, idle.value


def pool_trim(device=0):
    """hand every idle pooled block back to the driver (spc_pool_trim)"""
    _lib.call("spc_pool_trim", device)


class DeviceArray:
    """C-contiguous n-d array resident in HBM."""

    def __init__(self, shape, dtype, device=0, ptr=None, owner=None):
        self.shape = tuple(int(s) for s in np.atleast_1d(shape)) if not isinstance(shape, tuple) else tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.device = device
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self._owner = owner          # keeps a parent buffer alive for views
        self._owns = ptr is None
        if ptr is None:
            p = C.c_void_p()
            _lib.call("spc_malloc", device, max(self.nbytes, 1), C.byref(p))
            self.ptr = p.value
        else:
            self.ptr = int(ptr)

    # ---- construction / transfer ------------------------------------------
    @classmethod
    def from_numpy(cls, arr, device=0, stream=None, dtype=None):
        a = np.ascontiguousarray(arr, dtype=dtype)
        out = cls(a.shape, a.dtype, device)
        out.upload(a, stream)
        return out

    @classmethod
    def zeros(cls, shape, dtype, device=0, stream=None):
        out = cls(shape, dtype, device)
        _lib.call("spc_memset", device, C.c_void_p(out.ptr), 0, out.nbytes, _sh(stream))
        return out

    def upload(self, arr, stream=None):
        a = np.ascontiguousarray(arr, dtype=self.dtype)
        if a.nbytes != self.nbytes:
            raise ValueError("upload size mismatch: %d vs %d bytes" % (a.nbytes, self.nbytes))
        _lib.call("spc_memcpy_h2d", self.device, C.c_void_p(self.ptr), a.ctypes.data_as(C.c_void_p),
                  self.nbytes, _sh(stream))
        if stream is not None:
            _lib.call("spc_stream_sync", self.device, _sh(stream))   # `a` may be a temporary

    def get(self, stream=None):
        if getattr(self, "_is_view", False):
            raise ValueError("strided row view: copy through the parent array")
        out = None
        if _PinnedPool.MIN_BYTES <= self.nbytes <= _PinnedPool.MAX_BYTES:   # maps and small cubes; whole cubes stay pageable
            try:
                out = pinned_pool.array(self.shape, self.dtype)
            except (MemoryError, _lib.HipLibraryError):
                out = None
        if out is None:
            out = np.empty(self.shape, dtype=self.dtype)
        if stream is not None:
            _lib.call("spc_stream_sync", self.device, _sh(stream))
        if self.nbytes > _PinnedPool.MAX_BYTES and out.flags.c_contiguous:
            # the chunks come down on a NON-BLOCKING stream of their own, which is not ordered after the null stream the
            # way the synchronous copy below is: kernels launched with stream=None (every resident cube path) must have
            # finished before the first chunk leaves (hipStreamSynchronize(NULL) waits for the legacy default stream)
            if stream is None:
                _lib.call("spc_stream_sync", self.device, None)
            self._get_chunked(out)           # cube-sized: through pinned chunks, several host copies in flight
            return out
        _lib.call("spc_memcpy_d2h", self.device, out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr),
                  self.nbytes, None)
        return out

    def _get_chunked(self, out, chunk_bytes=64 << 20, nbuffers=8, workers=6):
        """Cube-sized results: one synchronous copy into fresh pageable memory runs at ~20 GB/s (the runtime stages it on
        one thread and the pages are touched for the first time).  Here 64 MiB chunks come down asynchronously into pinned
        buffers (the out-of-core pipeline's cache) and worker threads copy them into *out* - the first touch of the
        destination pages is spread over the workers - while the next chunks are in flight."""
        import ctypes
        from concurrent.futures import ThreadPoolExecutor
        from .streaming import _give_pinned, _take_pinned
        dst = out.reshape(-1).view(np.uint8)
        n = self.nbytes
        nchunks = (n + chunk_bytes - 1) // chunk_bytes
        nbuf = min(nbuffers, nchunks)
        pinned = _take_pinned(chunk_bytes, nbuf)
        stream = Stream(self.device)
        events, futs = [None] * nbuf, [None] * nbuf

        def land(k, ev, off, m):
            ev.synchronize()
            ctypes.memmove(dst[off:off + m].ctypes.data, pinned[k].ptr, m)

        try:
            with ThreadPoolExecutor(max_workers=workers) as pool:
                for i in range(nchunks):
                    k = i % nbuf
                    if futs[k] is not None:
                        futs[k].result()
                    off = i * chunk_bytes
                    m = min(chunk_bytes, n - off)
                    _lib.call("spc_memcpy_d2h", self.device, C.c_void_p(pinned[k].ptr), C.c_void_p(self.ptr + off), C.c_size_t(m),
                              stream.handle)
                    ev = Event(self.device)
                    ev.record(stream)
                    futs[k] = pool.submit(land, k, ev, off, m)
                for f in futs:
                    if f is not None:
                        f.result()
        finally:
            stream.synchronize()
            _give_pinned(chunk_bytes, pinned)

    def rows(self, y0, y1):
        """view of rows [y0, y1) of a (nz, ny, nx) array: same row/plane strides as the
        parent, no copy (what the C ABI's explicit strides are for)."""
        if len(self.shape) != 3:
            raise ValueError("rows() needs a (nz, ny, nx) array")
        nz, ny, nx = self.shape
        if not (0 <= y0 <= y1 <= ny):
            raise ValueError("row range out of bounds")
        v = DeviceArray((nz, y1 - y0, nx), self.dtype, self.device,
                        ptr=self.ptr + y0 * nx * self.dtype.itemsize, owner=self)
        v.row_stride = getattr(self, "row_stride", nx)
        v.plane_stride = getattr(self, "plane_stride", ny * nx)
        v.nbytes = 0            # not a contiguous buffer: upload()/get() are not available
        v._is_view = True
        return v

    def planes(self, z0, z1):
        """view of channels [z0, z1) of a (nz, ny, nx) array (no copy).  Of a contiguous parent it
        is itself contiguous; of a rows() view it keeps the parent's strides."""
        if len(self.shape) != 3:
            raise ValueError("planes() needs a (nz, ny, nx) array")
        nz, ny, nx = self.shape
        if not (0 <= z0 <= z1 <= nz):
            raise ValueError("channel range out of bounds")
        pstride = getattr(self, "plane_stride", ny * nx)
        v = DeviceArray((z1 - z0, ny, nx), self.dtype, self.device,
                        ptr=self.ptr + z0 * pstride * self.dtype.itemsize, owner=self)
        if getattr(self, "_is_view", False):
            v.row_stride, v.plane_stride = self.row_stride, self.plane_stride
            v.nbytes = 0
            v._is_view = True
        return v

    def swap01(self):
        """(ny, nz, nx) view of a (nz, ny, nx) array with the first two axes exchanged (no copy):
        row_stride and plane_stride trade places.  For the kernels that accept such a view
        (spc_percentile_axis0_f32: selection along y)."""
        if len(self.shape) != 3:
            raise ValueError("swap01() needs a (nz, ny, nx) array")
        nz, ny, nx = self.shape
        v = DeviceArray((ny, nz, nx), self.dtype, self.device, ptr=self.ptr, owner=self)
        v.row_stride = getattr(self, "plane_stride", ny * nx)
        v.plane_stride = getattr(self, "row_stride", nx)
        v.nbytes = 0
        v._is_view = True
        return v

    def reshape(self, shape):
        shape = tuple(int(s) for s in shape)
        if int(np.prod(shape, dtype=np.int64)) * self.dtype.itemsize != self.nbytes:
            raise ValueError("cannot reshape")
        return DeviceArray(shape, self.dtype, self.device, ptr=self.ptr, owner=self)

    def free(self):
        if self._owns and self.ptr:
            _lib.call("spc_free", self.device, C.c_void_p(self.ptr))
        self.ptr = 0
        self._owner = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def __repr__(self):
        return "DeviceArray(shape=%s, dtype=%s, device=%d, ptr=0x%x)" % (
            self.shape, self.dtype, self.device, self.ptr or 0)


def device_info(device=0):
    info = _lib.SpcDeviceInfo()
    _lib.call("spc_get_device_info", device, C.byref(info))
    return dict(name=info.name.decode(), arch=info.arch.decode(), compute_units=info.compute_units,
                wavefront_size=info.wavefront_size, total_mem=info.total_mem,
                free_mem=info.free_mem, clock_khz=info.clock_khz)
