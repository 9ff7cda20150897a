"""Chunk functions for DaskSpectralCube's apply-per-chunk seams.

The reference exposes two operator-level seams:

* ``DaskSpectralCubeMixin.apply_function_parallel_spectral(function,
  accepts_chunks=True)`` (spectral_cube/dask_spectral_cube.py:555-638): the
  function receives a NaN-filled ``(nz, cy, cx)`` numpy chunk with the whole
  spectral axis and returns an array of the same shape (or a reduced one when
  the caller passes ``drop_axis=[0]``);
* ``apply_function_parallel_spatial(function, accepts_chunks=True)``
  (:502-552): ``(cz, ny, nx)`` chunks with whole image planes.

Each factory below returns such a function.  The chunk is staged to HBM, the
HIP kernel runs, the result comes back as numpy - so the functions are
re-entrant (no global state; one stream and one pinned staging buffer per
worker thread) and picklable (module-level callables holding only numpy
arrays), which the dask ``threads`` / ``processes`` schedulers need
(dask_spectral_cube.py:278-312).  Masked voxels
arrive as NaN, so no mask is passed to the kernels.  Empty chunks are passed
through like the reference's wrappers do (:600-610).
"""
import ctypes as C
import os
import threading

import numpy as np

from . import _lib, ops
from .device import DeviceArray, Stream, pinned_pool, _PinnedPool

# ---- PCIe staging of a chunk (round 3) ------------------------------------------------------------------
# Every worker thread of dask's `threads` scheduler owns ONE stream and ONE page-locked input buffer (grown to
# the largest chunk it has seen).  A call copies (and converts) the chunk into the pinned buffer, queues H2D ->
# kernel -> D2H on the thread's stream and waits for that stream only - never for the device - so the H2D, the
# kernel and the D2H of chunks handled by neighbouring threads overlap.  Results land in page-locked numpy arrays
# from device.pinned_pool (handed to dask as they are: no extra host copy; the buffer returns to the pool when
# dask drops the array).  Round 2 staged from pageable memory with synchronous copies (3 - 16 GB/s and a device
# drain per copy).
_tls = threading.local()


class _ThreadStage:
    def __init__(self, device):
        self.device = device
        self.stream = Stream(device)
        self.ptr, self.cap = 0, 0
        self.dev = {}               # tag -> DeviceArray reused from chunk to chunk

    def buffer(self, tag, shape, dtype):
        """the thread's device buffer for *tag*: chunks of one graph have (nearly) all the same shape, and giving a block
        back to the library's pool drains the DEVICE (spc_free: nothing queued may still use it) - with 8 worker threads
        freeing an input and an output per chunk the workers end up waiting for each other's kernels"""
        shape = tuple(int(s) for s in shape)
        a = self.dev.get(tag)
        if a is None or a.shape != shape or a.dtype != np.dtype(dtype):
            self.stream.synchronize()
            a = self.dev[tag] = DeviceArray(shape, dtype, self.device)
        return a

    def pinned(self, nbytes):
        if nbytes > self.cap:
            if self.ptr:
                self.stream.synchronize()
                _lib.call("spc_host_free", C.c_void_p(self.ptr))
                self.ptr, self.cap = 0, 0
            cap = 1 << max(int(nbytes) - 1, 1 << 20).bit_length()
            p = C.c_void_p()
            _lib.call("spc_host_alloc", C.c_size_t(cap), C.byref(p))
            self.ptr, self.cap = p.value, cap
        return self.ptr

    def __del__(self):
        try:
            if self.ptr:
                _lib.call("spc_host_free", C.c_void_p(self.ptr))
        except Exception:
            pass



def _host_result_budget():
    """bytes a cube -> cube result may take in host RAM: half of MemAvailable (SPC_DASK_RESULT_RAM_MB overrides)"""
    env = os.environ.get("SPC_DASK_RESULT_RAM_MB")
    if env:
        return int(env) << 20
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    return int(line.split()[1]) * 1024 // 2
    except OSError:
        pass
    return 1 << 62


def _unlink_quiet(path):
    try:
        os.unlink(path)
    except OSError:
        pass


def _ctx(device):
    stages = getattr(_tls, "stages", None)
    if stages is None:
        stages = _tls.stages = {}
    st = stages.get(device)
    if st is None:
        st = stages[device] = _ThreadStage(device)
    return st


_helpers = None
_helpers_lock = threading.Lock()


def _copy_pool():
    """threads that copy pieces of a chunk into the pinned buffers (numpy releases the GIL for these copies).  A dask chunk
    of apply_function_parallel_spectral is a (nz, cy, cx) window of the cube - rows of cx samples, a KiB or so apart in
    memory - and one thread moves such a window at 3 - 5 GB/s: with one copy per worker the link idles."""
    global _helpers
    with _helpers_lock:
        if _helpers is None:
            from concurrent.futures import ThreadPoolExecutor
            import os
            try:
                ncpu = len(os.sched_getaffinity(0))
            except AttributeError:
                ncpu = os.cpu_count() or 4
            _helpers = ThreadPoolExecutor(max_workers=max(2, min(32, ncpu // 2)), thread_name_prefix="spc-stage")
    return _helpers


def _stage(chunk, device):
    """chunk (any real dtype, any strides) -> float32 DeviceArray, through the thread's pinned buffer and stream.
    Returns (DeviceArray, thread stage); the copy is queued, not awaited.  The chunk is cut into runs of planes: helper
    threads convert / copy them into the pinned buffer side by side, and every run goes up as soon as it has been copied -
    the H2D of run k overlaps the host copy of run k + 1."""
    st = _ctx(device)
    n = int(chunk.size)
    st.stream.synchronize()                                  # the previous chunk's H2D has left the buffer
    ptr = st.pinned(n * 4)
    view = np.frombuffer((C.c_byte * (n * 4)).from_address(ptr), dtype=np.float32, count=n).reshape(chunk.shape)
    dev = st.buffer("in", chunk.shape, np.float32)
    nz = chunk.shape[0] if chunk.ndim == 3 else 0
    plane = n // nz if nz else 0
    pieces = min(8, nz, (n * 4) >> 24) if nz else 0          # runs of >= 16 MiB
    if pieces < 2:
        np.copyto(view, chunk, casting="unsafe")
        _lib.call("spc_memcpy_h2d", device, C.c_void_p(dev.ptr), C.c_void_p(ptr), C.c_size_t(n * 4), st.stream.handle)
        return dev, st
    pool = _copy_pool()
    cuts = [nz * k // pieces for k in range(pieces + 1)]
    jobs = [pool.submit(np.copyto, view[a:b], chunk[a:b], casting="unsafe") for a, b in zip(cuts[:-1], cuts[1:])]
    for (a, b), job in zip(zip(cuts[:-1], cuts[1:]), jobs):
        job.result()
        off = a * plane * 4
        _lib.call("spc_memcpy_h2d", device, C.c_void_p(dev.ptr + off), C.c_void_p(ptr + off), C.c_size_t((b - a) * plane * 4),
                  st.stream.handle)
    return dev, st


def _fetch(dev, st, dtype=None):
    """DeviceArray -> numpy array (page-locked when the pool has room), on the thread's stream; waits for it"""
    out = None
    if _PinnedPool.MIN_BYTES <= dev.nbytes <= _PinnedPool.MAX_BYTES:
        try:
            out = pinned_pool.array(dev.shape, dev.dtype)
        except (MemoryError, _lib.HipLibraryError):
            out = None
    if out is None:
        out = np.empty(dev.shape, dtype=dev.dtype)
    _lib.call("spc_memcpy_d2h", dev.device, out.ctypes.data_as(C.c_void_p), C.c_void_p(dev.ptr), C.c_size_t(dev.nbytes),
              st.stream.handle)
    st.stream.synchronize()
    return out if dtype is None else out.astype(dtype, copy=False)


class SpectralSmoothChunk:
    """drop-in for the ``spectral_smooth`` chunk function
    (dask_spectral_cube.py:912-914)."""

    def __init__(self, kernel, device=0):
        self.kernel = np.asarray(getattr(kernel, "array", kernel), dtype=np.float64)
        self.device = device

    def __call__(self, chunk):
        if chunk.size == 0:
            return chunk
        dev, st = _stage(chunk, self.device)
        return _fetch(ops.spectral_conv(dev, self.kernel, stream=st.stream, out=st.buffer("out", chunk.shape, np.float32)), st, chunk.dtype)


class SpatialSmoothChunk:
    """drop-in for the ``spatial_smooth`` chunk function (:990-993, :540-547)."""

    def __init__(self, kernel, device=0):
        self.kernel = np.asarray(getattr(kernel, "array", kernel), dtype=np.float64)
        self.device = device

    def __call__(self, chunk, **kwargs):
        if chunk.size == 0:
            return chunk
        dev, st = _stage(chunk, self.device)
        return _fetch(ops.spatial_conv(dev, self.kernel, stream=st.stream, out=st.buffer("out", chunk.shape, np.float32)), st, chunk.dtype)


class SigmaClipChunk:
    """drop-in for the chunk function of ``sigma_clip_spectrally`` (dask_spectral_cube.py:851-878: astropy's
    ``sigma_clip(chunk, sigma=threshold, axis=0, masked=False, **kwargs)`` under
    ``apply_function_parallel_spectral(accepts_chunks=True)`` - the operation docs/dask.rst:176-275 times): the whole
    clip loop of a chunk is one kernel with the rays resident in registers."""

    def __init__(self, threshold, device=0, **kwargs):
        unknown = set(kwargs) - {"sigma_lower", "sigma_upper", "maxiters", "cenfunc", "stdfunc"}
        if unknown:
            raise NotImplementedError("sigma_clip options not on the device path: %s" % sorted(unknown))
        self.threshold, self.kwargs, self.device = float(threshold), kwargs, device

    def __call__(self, chunk, **ignored):
        if chunk.size == 0:
            return chunk
        dev, st = _stage(chunk, self.device)
        return _fetch(ops.sigma_clip_axis0(dev, sigma=self.threshold, stream=st.stream, **self.kwargs), st, chunk.dtype)


class MomentChunk:
    """reduced chunk function (use with ``drop_axis=[0]``): moment map of a
    ``(nz, cy, cx)`` chunk.  ``pix_cen`` = offsets from channel 0, ``pix_size``
    and ``world0`` as in dask_spectral_cube.py:1083-1123."""

    def __init__(self, order, pix_cen, pix_size, world0=0.0, device=0):
        if order not in (0, 1, 2):
            raise ValueError("MomentChunk supports order 0, 1, 2")
        self.order = order
        self.pix_cen = np.asarray(pix_cen, dtype=np.float64)
        self.pix_size = float(pix_size)
        self.world0 = float(world0)
        self.device = device

    def __call__(self, chunk):
        if chunk.size == 0:
            return chunk.sum(axis=0)
        nz = chunk.shape[0]
        cref = self.pix_cen[nz // 2]
        key = ("m0", "m1", "m2")[self.order]
        dev, st = _stage(chunk, self.device)
        cen = st.buffer("cen", (nz,), np.float64)
        cen.upload(self.pix_cen - cref, st.stream)
        out = {key: st.buffer("map", chunk.shape[1:], np.float64)}
        ws = st.buffer("ws", (max(1, int(_lib.load().spc_moments_workspace_bytes(*chunk.shape))),), np.uint8)
        ops.moments(dev, cen, dv=self.pix_size, m1_add=cref + self.world0, want=(key,), stream=st.stream, out=out, workspace=ws)
        return _fetch(out[key], st)


class Moments012Chunk:
    """moment 0, 1 AND 2 of a ``(nz, cy, cx)`` chunk from ONE staging and ONE launch: a ``(3, cy, cx)`` float64 block
    (use with ``drop_axis=[0], new_axis=[0], chunks=((3,), cy_chunks, cx_chunks)``).  The reference runs three graph
    executions for order 2 (dask_spectral_cube.py:1090,1097,1104) and round 2's MomentChunk staged the chunk once
    per order; the chunk crosses PCIe once here."""

    def __init__(self, pix_cen, pix_size, world0=0.0, device=0):
        self.pix_cen = np.asarray(pix_cen, dtype=np.float64)
        self.pix_size = float(pix_size)
        self.world0 = float(world0)
        self.device = device

    def __call__(self, chunk):
        if chunk.size == 0:
            return np.zeros((3,) + chunk.shape[1:], np.float64)
        nz, cy, cx = chunk.shape
        cref = self.pix_cen[nz // 2]
        dev, st = _stage(chunk, self.device)
        block = st.buffer("maps", (3, cy, cx), np.float64)
        out = {k: DeviceArray((cy, cx), np.float64, self.device, ptr=block.ptr + i * cy * cx * 8, owner=block)
               for i, k in enumerate(("m0", "m1", "m2"))}
        cen = st.buffer("cen", (nz,), np.float64)
        cen.upload(self.pix_cen - cref, st.stream)
        ws = st.buffer("ws", (max(1, int(_lib.load().spc_moments_workspace_bytes(nz, cy, cx))),), np.uint8)
        ops.moments(dev, cen, dv=self.pix_size, m1_add=cref + self.world0, want=("m0", "m1", "m2"), stream=st.stream, out=out,
                    workspace=ws)
        return _fetch(block, st)


class SpectralInterpolateChunk:
    """drop-in for ``interp_wrapper`` (dask_spectral_cube.py:1342-1353)."""

    def __init__(self, inaxis, grid, fill_value=None, device=0):
        self.plan = ops.lerp_plan(inaxis, grid, fill_value)
        if self.plan[3] or self.plan[4]:
            raise ValueError("pass ascending axes; the caller flips the data like the reference does")
        self.device = device

    def __call__(self, chunk):
        if chunk.size <= 1:
            return chunk
        lo, t, inv, _, _, fill = self.plan
        dev, st = _stage(chunk, self.device)
        out = st.buffer("out", (len(lo),) + tuple(chunk.shape[1:]), np.float32)
        return _fetch(ops.spectral_lerp(dev, lo, t, inv, fill, stream=st.stream, out=out), st)


# ---- one level higher: the cube-level operators of a dask-backed cube (round 4) ------------------------------------------
def _prefault(arr, threads=8):
    """touch every page of a fresh result array from several threads (4 GiB: 45 ms with 8 threads against 260 ms when the
    writer threads of the sink meet the faults one page at a time)"""
    v = arr.view(np.uint8).reshape(-1)
    if v.size < (64 << 20):
        return
    step = -(-v.size // threads)
    pool = _copy_pool()
    jobs = [pool.submit(v[i * step:(i + 1) * step:4096].fill, 0) for i in range(threads)]
    for j in jobs:
        j.result()


class DaskCubeOps:
    """``DaskSpectralCubeMixin``-shaped operators for a cube whose data are a dask array, WITHOUT the per-chunk seam:
    ``moment`` / ``moments012`` / ``spectral_smooth`` / ``spatial_smooth`` / ``sigma_clip_spectrally`` read windows of the
    dask array straight into the pinned ring of the out-of-core strip pipeline (streaming.DaskSource + Strips) and run
    the strip kernels; cube -> cube results come back as a dask array over the host sink the strips were written to.
    This is what a ``DaskSpectralCube`` method override calls:

        ops = DaskCubeOps(self._get_filled_data(fill=np.nan), self.header)      # NaN-filled data: dask_spectral_cube.py:205-230
        return self._new_cube_with(data=ops.spectral_smooth(kernel), ...)        # :836-840 keeps the mask

    The per-chunk functions above remain for ``apply_function_parallel_*`` with user functions (:502-638)."""

    def __init__(self, data, header, device=0, chunks=None):
        from .cube import SpectralCube
        from . import masks as M, streaming
        self._shape = tuple(int(s) for s in data.shape)
        self._chunks = chunks if chunks is not None else getattr(data, "chunks", None)
        src = streaming.DaskSource(data)
        self.cube = SpectralCube(None, header=header, device=device, _source=src, _shape=self._shape)
        self.cube._mask = M.LazyMask(np.isfinite, cube=self.cube)         # masked voxels arrive as NaN (FilledArrayHandler)

    def moment(self, order=0, axis=0):
        return np.asarray(self.cube.moment(order=order, axis=axis))

    def moments012(self):
        return tuple(np.asarray(m) for m in self.cube.moments012())

    def _to_dask(self, pending):
        import dask.array as da
        shape = tuple(pending.shape)
        nbytes = 4 * int(np.prod(shape))
        # The reference keeps these results lazy; here calling the method already runs the strip pipeline, and the
        # result has to land somewhere.  In RAM while it fits beside what else lives there (half of MemAvailable);
        # beyond that in a memory-mapped .npy under SPC_DASK_SPILL_DIR (default: the temp dir), so a cube larger than
        # host memory stays out of core instead of ending in the OOM killer.
        if nbytes > _host_result_budget():
            import tempfile, uuid as _uuid
            d = os.environ.get("SPC_DASK_SPILL_DIR") or tempfile.gettempdir()
            path = os.path.join(d, "spc-dask-result-%d-%s.npy" % (os.getpid(), _uuid.uuid4().hex))
            out = np.lib.format.open_memmap(path, mode="w+", dtype=np.float32, shape=shape)
            import weakref
            weakref.finalize(out, _unlink_quiet, path)
        else:
            out = np.empty(shape, dtype=np.float32)
            _prefault(out)
        pending.stream_into(out)
        chunks = self._chunks if (self._chunks is not None and tuple(pending.shape) == self._shape) else "auto"
        # (an explicit name: by default from_array HASHES the array's contents for its task name - 1.5 s for 4 GiB)
        import uuid
        return da.from_array(out, chunks=chunks, name="spc-result-" + uuid.uuid4().hex)

    def spectral_smooth(self, kernel):
        return self._to_dask(self.cube.spectral_smooth(kernel))

    def spatial_smooth(self, kernel, **kwargs):
        return self._to_dask(self.cube.spatial_smooth(kernel, **kwargs))

    def sigma_clip_spectrally(self, threshold, **kwargs):
        return self._to_dask(self.cube.sigma_clip_spectrally(threshold, **kwargs))
