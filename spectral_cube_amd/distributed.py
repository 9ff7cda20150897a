"""Multi-GPU driver: (y, x) row-strip sharding + one all-gather of the 2-D map.

Every row of SURVEY.md section 8(a) is independent per spaxel, so a cube
shards over contiguous row strips ``(nz, ny/G, nx)`` - one process per GPU, no
collective on the data path.  The only exchange is the stitch of the final
2-D map strips: ONE all-gather (RCCL over xGMI on GPUs; a gloo all-gather of
host arrays in the CPU tests).  The reference has no counterpart (its
parallelism is dask chunking, dask_spectral_cube.py:259-312).

torch.distributed is used here only as the rendezvous / bootstrap plumbing the
launcher (`python -m torch.distributed.run`) already provides; the RCCL calls
themselves go through the C ABI (spc_comm_init / spc_allgather_rows).
"""
import ctypes as C

import numpy as np

from . import _lib
from .device import DeviceArray, _sh


def strip_bounds(ny, world_size, rank):
    """Rows [y0, y1) owned by *rank*: equal strips of ceil(ny/world) rows, the
    last ones possibly short/empty (the gather pads them)."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank %d / world %d" % (rank, world_size))
    rows = -(-ny // world_size)
    y0 = min(ny, rank * rows)
    y1 = min(ny, y0 + rows)
    return y0, y1


def strip_rows(ny, world_size):
    return -(-ny // world_size)


def halo_bounds(ny, world_size, rank, halo):
    """Rows a rank must hold to smooth its strip spatially: the strip plus *halo* rows on
    each side (clipped at the cube's edges).  Returns (h0, h1, top, nrows): the rank loads
    rows [h0, h1), its own strip is rows [top, top + nrows) of that extended strip
    (halo = kernel half width, 14 rows for the 29-tap config C4)."""
    y0, y1 = strip_bounds(ny, world_size, rank)
    h0, h1 = max(0, y0 - halo), min(ny, y1 + halo)
    return h0, h1, y0 - h0, y1 - y0


def smooth_moment0_strip(ext_cube, kernel2d, top, nrows, mask=None, dv=1.0, stream=None):
    """spatial_smooth + moment0 of one rank's extended strip, all on the device.

    ext_cube: (nz, h1-h0, nx) float32 DeviceArray holding the strip and its halo rows;
    mask: MaskSpec for the same extended strip (or None).  The smoothed halo rows are
    wrong (they see an artificial boundary) and are simply not reduced: the moment kernel
    reads rows [top, top+nrows) in place through strides.  Returns the (nrows, nx)
    float64 moment-0 strip (DeviceArray) - the only thing that is all-gathered.
    The smoothed cube keeps the ORIGINAL mask (dask_spectral_cube.py:836-840)."""
    from . import ops
    sm = ops.spatial_conv(ext_cube, kernel2d, mask=mask, stream=stream)
    nz = ext_cube.shape[0]
    cen = DeviceArray.zeros((nz,), np.float64, ext_cube.device)
    r = ops.moments(sm.rows(top, top + nrows), cen, dv=dv, want=("m0",), stream=stream,
                    mask=mask.rows(top, top + nrows) if mask is not None else None)
    r["_smoothed"] = sm
    return r["m0"]


def read_strip(path, rank, world_size, halo=0, device=None, **kw):
    """Each rank streams ONLY its row strip (plus *halo* rows per side) of a FITS cube into its
    own GPU (io_fits.load_cube(rows=...)): the reader feeding (y, x) tiles of SURVEY.md
    section 8e/8f.  Returns (DeviceArray, header, top, nrows) with the rank's own rows at
    [top, top + nrows) of the loaded strip."""
    from . import io_fits
    img = io_fits.find_image(path, kw.get("hdu"))
    ny = io_fits.cube_shape(img)[1]
    h0, h1, top, n = halo_bounds(ny, world_size, rank, halo)
    dev, hdr = io_fits.load_cube(path, device=rank if device is None else device, rows=(h0, h1), **kw)
    return dev, hdr, top, n


class HostGatherComm:
    """all-gather of host strips through torch.distributed (gloo).  Used by the
    world_size-2 CPU tests and as the loud, explicitly reported stitch fallback
    of bench.py when RCCL cannot initialise."""

    kind = "gloo-host"

    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)

    def allgather_rows(self, strip, ny_total):
        import torch
        strip = np.ascontiguousarray(strip)
        rows = strip_rows(ny_total, self.world_size)
        pad = np.full((rows,) + strip.shape[1:], np.nan if strip.dtype.kind == "f" else 0, dtype=strip.dtype)
        pad[:strip.shape[0]] = strip
        t = torch.from_numpy(pad)
        outs = [torch.empty_like(t) for _ in range(self.world_size)]
        self._dist.all_gather(outs, t, group=self.group)
        full = np.concatenate([o.numpy() for o in outs], axis=0)
        return full[:ny_total]

    def barrier(self):
        self._dist.barrier(group=self.group)


class RcclComm:
    """device all-gather through libspcube_hip.so -> RCCL (xGMI)."""

    kind = "rccl"

    def __init__(self, device, rank, world_size, bcast_bytes):
        """bcast_bytes(payload_or_None) -> bytes: broadcast rank 0's payload to
        all ranks (any host mechanism: torch.distributed, a shared file...)."""
        self.device, self.rank, self.world_size = device, rank, world_size
        ident = None
        if rank == 0:
            buf = (C.c_uint8 * _lib.COMM_ID_BYTES)()
            _lib.call("spc_comm_unique_id", buf)
            ident = bytes(buf)
        ident = bcast_bytes(ident)
        buf = (C.c_uint8 * _lib.COMM_ID_BYTES).from_buffer_copy(ident)
        h = C.c_void_p()
        _lib.call("spc_comm_init", device, buf, world_size, rank, C.byref(h))
        self._h = h

    def allgather_rows_device(self, strip_dev, recv_dev, stream=None):
        """strip_dev: (rows, nx) DeviceArray, identical shape on every rank;
        recv_dev: (world*rows, nx)."""
        if recv_dev.nbytes != strip_dev.nbytes * self.world_size:
            raise ValueError("receive buffer must hold world_size strips")
        _lib.call("spc_allgather_rows", self._h, _sh(stream), C.c_void_p(strip_dev.ptr),
                  C.c_void_p(recv_dev.ptr), strip_dev.nbytes)
        return recv_dev

    def allgather_rows(self, strip_dev, ny_total, stream=None):
        rows = strip_rows(ny_total, self.world_size)
        if strip_dev.shape[0] != rows:
            raise ValueError("RcclComm needs equal strips (pad the last rank to %d rows)" % rows)
        recv = DeviceArray((rows * self.world_size,) + tuple(strip_dev.shape[1:]), strip_dev.dtype, self.device)
        self.allgather_rows_device(strip_dev, recv, stream)
        return recv

    def close(self):
        if getattr(self, "_h", None):
            _lib.call("spc_comm_destroy", self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def torch_bcast_bytes(group=None):
    """bootstrap helper: broadcast bytes from rank 0 over torch.distributed."""
    import torch.distributed as dist

    def bcast(payload):
        obj = [payload]
        dist.broadcast_object_list(obj, src=0, group=group)
        return obj[0]
    return bcast


def sharded_moments(strip_cube, ny_total, comm, orders=(0, 1, 2)):
    """moment maps of a cube sharded by row strips: every rank computes the
    maps of ITS strip with the fused HIP kernel, then one all-gather stitches
    them.  Returns {order: (ny_total, nx) float64 ndarray} on every rank."""
    keys = {0: "m0", 1: "m1", 2: "m2"}
    want = tuple(keys[o] for o in orders)
    r = strip_cube._moment_device(want)
    out = {}
    for o in orders:
        strip = r[keys[o]]
        if isinstance(comm, RcclComm):
            rows = strip_rows(ny_total, comm.world_size)
            if strip.shape[0] != rows:       # pad a short last strip on the host
                padded = np.full((rows, strip.shape[1]), np.nan)
                padded[:strip.shape[0]] = strip.get()
                strip = DeviceArray.from_numpy(padded, strip.device)
            out[o] = comm.allgather_rows(strip, ny_total).get()[:ny_total]
        else:
            out[o] = comm.allgather_rows(strip.get() if isinstance(strip, DeviceArray) else strip, ny_total)
    return out


def combine_statistics(parts):
    """statistics() of a cube from the per-strip records {npts, min, max, sum, sumsq} (the aggregation step
    of DaskSpectralCubeMixin.statistics, dask_spectral_cube.py:795-814, across ranks instead of chunks)."""
    npts = sum(int(p["npts"]) for p in parts)
    live = [p for p in parts if int(p["npts"]) > 0]
    out = {"npts": npts,
           "min": min((float(p["min"]) for p in live), default=float("nan")),
           "max": max((float(p["max"]) for p in live), default=float("nan")),
           "sum": float(sum(float(p["sum"]) for p in live)), "sumsq": float(sum(float(p["sumsq"]) for p in live))}
    with np.errstate(invalid="ignore", divide="ignore"):      # same formulae as SpectralCube.statistics
        out["mean"] = out["sum"] / npts if npts else float("nan")
        out["sigma"] = float(np.sqrt((out["sumsq"] - out["sum"] ** 2 / npts) / (npts - 1))) if npts > 1 else float("nan")
        out["rms"] = float(np.sqrt(out["sumsq"] / npts)) if npts else float("nan")
    return out


def sharded_statistics(strip_stats, group=None):
    """statistics() of a row-sharded cube: every rank passes the record of ITS strip (ops.stats_global on
    the device, one pass), five numbers per rank travel through torch.distributed.all_gather_object."""
    import torch.distributed as dist
    parts = [None] * dist.get_world_size(group)
    dist.all_gather_object(parts, {k: float(strip_stats[k]) for k in ("npts", "min", "max", "sum", "sumsq")}, group=group)
    return combine_statistics(parts)


def sharded_smooth_moment0(strip_cube, kernel, ny_total, comm):
    """config C4 (spatial_smooth -> moment0) on row strips WITHOUT halo exchange: when every voxel
    is valid, smoothing commutes with the sum along the spectral axis, so each rank reduces its
    own strip (sum map S0 + valid count), ONE all-gather stitches S0, and the 2-D convolution
    runs on the stitched map (spc_map_conv2d_f64).  A rank that holds an invalid voxel poisons
    its S0 strip with NaN, which every rank then sees in the stitched map: returns None
    everywhere -> fall back to halo_bounds / smooth_moment0_strip on extended strips.
    Returns the (ny_total, nx) float64 moment-0 map (ndarray) on every rank."""
    from . import ops
    from .kernels import kernel_array
    nz = strip_cube.shape[0]
    r = strip_cube._moment_device(("s0", "nvalid"))
    s0 = r["s0"]
    if int(r["nvalid"].get().min()) != nz:
        s0 = DeviceArray.from_numpy(np.full(s0.shape, np.nan), s0.device)
    if isinstance(comm, RcclComm):
        rows = strip_rows(ny_total, comm.world_size)
        if s0.shape[0] != rows:
            padded = np.zeros((rows, s0.shape[1]))
            padded[:s0.shape[0]] = s0.get()
            s0 = DeviceArray.from_numpy(padded, s0.device)
        full = comm.allgather_rows(s0, ny_total).get()[:ny_total]
    else:
        full = comm.allgather_rows(s0.get(), ny_total)
    if not np.all(np.isfinite(full)):
        return None
    karr = kernel_array(kernel, 2)
    c0 = ops.map_conv2d(DeviceArray.from_numpy(np.ascontiguousarray(full), strip_cube.device), karr).get()
    return strip_cube._pix_size_slice(0) * c0
