"""Multi-GPU driver: (y, x) row-strip sharding + one all-gather of the 2-D map.

Every row of SURVEY.md section 8(a) is independent per spaxel, so a cube
shards over contiguous row strips ``(nz, ny/G, nx)`` - one process per GPU, no
collective on the data path.  The only exchange is the stitch of the final
2-D map strips: ONE all-gather (RCCL over xGMI on GPUs; an all-gather of host
arrays through the rendezvous transport in the CPU tests).  The reference has no
counterpart (its parallelism is dask chunking, dask_spectral_cube.py:259-312).

No torch here: the 128-byte RCCL id and the few host-side records travel through a
*transport* object (``rendezvous.FileRendezvous`` in production: files in a shared
directory; the CPU tests also run the drivers over a gloo transport of their own).
The RCCL calls go through the C ABI (spc_comm_init / spc_allgather_rows).

Drivers, by BASELINE config:
  C2 / C4 maps      sharded_moments                 strip moments + one all-gather
  C3                sharded_spectral_smooth_moment  fused smooth->moment per strip + one all-gather
  C4                sharded_smooth_moment0 (no halo, all-valid) / smooth_moment0_strip (halo rows)
  C5                sharded_spectral_interpolate    per-spaxel, no exchange
                    reproject_source_rows + sharded_reproject   output-row strips; every rank loads
                                                    only the source rows its strip's pixel map touches
  statistics        sharded_statistics              five numbers per rank
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
    """all-gather of host strips through a rendezvous transport (rendezvous.FileRendezvous, or the
    gloo transport of the world_size-2 CPU tests).  Also the loud, explicitly reported stitch
    fallback of bench.py when RCCL cannot initialise."""

    kind = "host"

    def __init__(self, transport):
        self.transport = transport
        self.rank, self.world_size = transport.rank, transport.world_size

    def allgather_rows(self, strip, ny_total):
        strip = np.ascontiguousarray(strip)
        rows = strip_rows(ny_total, self.world_size)
        pad = np.full((rows,) + strip.shape[1:], np.nan if strip.dtype.kind == "f" else 0, dtype=strip.dtype)
        pad[:strip.shape[0]] = strip
        parts = self.transport.allgather_bytes(pad.tobytes())
        full = np.concatenate([np.frombuffer(b, dtype=strip.dtype).reshape(pad.shape) for b in parts], axis=0)
        return full[:ny_total]

    def barrier(self):
        self.transport.barrier()


class RcclComm:
    """device all-gather through libspcube_hip.so -> RCCL (xGMI)."""

    kind = "rccl"

    def __init__(self, device, transport):
        """transport: rendezvous object (rank, world_size, bcast_bytes): carries rank 0's unique id."""
        self.device, self.rank, self.world_size = device, transport.rank, transport.world_size
        ident, err = None, None
        if self.rank == 0:
            try:
                buf = (C.c_uint8 * _lib.COMM_ID_BYTES)()
                _lib.call("spc_comm_unique_id", buf)
                ident = bytes(buf)
            except Exception as exc:          # the broadcast still happens (an empty id): every rank stays on the
                ident, err = b"", exc         # same rendezvous sequence number and fails here TOGETHER
        ident = transport.bcast_bytes(ident)
        if len(ident) != _lib.COMM_ID_BYTES:
            raise _lib.HipLibraryError("rank 0 could not create an RCCL id%s" % (": %s" % err if err else ""))
        buf = (C.c_uint8 * _lib.COMM_ID_BYTES).from_buffer_copy(ident)
        h = C.c_void_p()
        _lib.call("spc_comm_init", device, buf, self.world_size, self.rank, C.byref(h))
        self._h = h

    def allgather_rows_device(self, strip_dev, recv_dev, stream=None):
        """strip_dev: (rows, nx) DeviceArray, identical shape on every rank;
        recv_dev: (world*rows, nx)."""
        if recv_dev.nbytes != strip_dev.nbytes * self.world_size:
            raise ValueError("receive buffer must hold world_size strips")
        _lib.call("spc_allgather_rows", self._h, _sh(stream), C.c_void_p(strip_dev.ptr),
                  C.c_void_p(recv_dev.ptr), strip_dev.nbytes)
        return recv_dev

    def allgather_batch(self, pairs, stream=None):
        """several all-gathers in ONE grouped RCCL launch.  pairs: [(send DeviceArray, (destination pointer, bytes the
        destination holds))] - every send has the same shape on every rank, its destination world_size times its size."""
        n = len(pairs)
        sends = (C.c_void_p * n)(*[p[0].ptr for p in pairs])
        recvs = (C.c_void_p * n)(*[p[1][0] for p in pairs])
        sizes = (C.c_size_t * n)(*[p[0].nbytes for p in pairs])
        for send, (_, room) in pairs:
            if room < send.nbytes * self.world_size:
                raise ValueError("receive region must hold world_size strips")
        _lib.call("spc_allgather_rows_batch", self._h, _sh(stream), n, sends, recvs, sizes)

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


class ChunkedMoments:
    """moment 0 / 1 / 2 of a row-sharded cube with the stitch hidden INSIDE the call: the rank's rows are taken in
    `chunks` blocks; the kernel of block c+1 runs while the maps of block c are all-gathered on a second stream (one
    grouped RCCL launch for the three maps of a block).  Row ownership is block-cyclic - with `rc` rows per block, block c
    of rank r holds the rows c * (ny_total / chunks) + r * rc + [0, rc) of the cube - so that every all-gather lands its
    world_size blocks in natural row order and the three (ny_total, nx) maps need no re-layout.  The latency of one call is
    the kernel time plus the all-gather of the LAST block, instead of kernel + whole all-gather.
    cube / mask_array: this rank's (nz, chunks * rc, nx) DeviceArrays (local row block c = cube rows [c * rc, (c+1) * rc))."""

    KEYS = ("m0", "m1", "m2")

    @staticmethod
    def pick_chunks(rows, min_block_rows=128, max_chunks=4):
        """blocks per call from the strip height: the largest count <= max_chunks that divides the rows and keeps every
        block at min_block_rows or more - the headline kernel loses 6 % on 128-row blocks of 2048 columns and 19 % on 64-row
        ones - so 4 blocks on strips of 512 rows and more (N <= 4 of the north-star cube), 2 at N = 8,
        1 (no overlap, nothing lost) below 256 rows."""
        for c in range(max_chunks, 1, -1):
            if rows % c == 0 and rows // c >= min_block_rows:
                return c
        return 1

    def __init__(self, cube, mask_array, d_cen, dv, m1_add, comm, chunks=None, workspace=None, want=("m0", "m1", "m2")):
        from . import ops
        self.ops, self.comm = ops, comm
        nz, rows, nx = cube.shape
        if chunks is None:
            chunks = self.pick_chunks(rows)
        if rows % chunks:
            raise ValueError("the rank's %d rows do not split into %d chunks" % (rows, chunks))
        want = tuple(k for k in self.KEYS if k in want)      # only the orders asked for are computed, sent and stitched
        if not want:
            raise ValueError("want must name at least one of %r" % (self.KEYS,))
        self.want = want
        self.chunks, self.rc, self.nx = chunks, rows // chunks, nx
        self.ny_total = rows * comm.world_size
        dev = cube.device
        self.args = dict(dv=dv, m1_add=m1_add, want=want, workspace=workspace)
        self.d_cen = d_cen
        rc = self.rc
        self.cubes = [cube.rows(c * rc, (c + 1) * rc) for c in range(chunks)]
        self.masks = [None if mask_array is None else ops.MaskSpec(_lib.MASK_ARRAY, array=mask_array.rows(c * rc, (c + 1) * rc))
                      for c in range(chunks)]
        self.sends = [{k: DeviceArray((rc, nx), np.float64, dev) for k in want} for _ in range(chunks)]
        self.maps = {k: DeviceArray((self.ny_total, nx), np.float64, dev) for k in want}
        from .device import Event
        self.ev = [Event(dev) for _ in range(chunks)]
        self.ev_done = Event(dev)

    @property
    def stitch_bytes_per_rank(self):
        """bytes one rank sends per call"""
        return len(self.want) * self.chunks * self.rc * self.nx * 8

    def __call__(self, stream, comm_stream):
        """enqueue one call; the maps are complete when `stream` has drained (it waits for the last all-gather)."""
        block = self.rc * self.nx * 8
        for c in range(self.chunks):
            self.ops.moments(self.cubes[c], self.d_cen, mask=self.masks[c], stream=stream, out=self.sends[c], **self.args)
            self.ev[c].record(stream)
            comm_stream.wait_event(self.ev[c])
            base = c * self.comm.world_size * block
            self.comm.allgather_batch([(self.sends[c][k], (self.maps[k].ptr + base, self.comm.world_size * block))
                                       for k in self.want], comm_stream)
        self.ev_done.record(comm_stream)
        stream.wait_event(self.ev_done)
        return self.maps

    def global_rows(self, rank, c):
        """rows of the cube that block c of `rank` holds"""
        y0 = c * (self.ny_total // self.chunks) + rank * self.rc
        return y0, y0 + self.rc


def _stitch(strip, ny_total, comm, pad_value=np.nan):
    """(ny_total, nx) ndarray on every rank from the ranks' (rows, nx) map strips: ONE all-gather."""
    if isinstance(comm, RcclComm):
        rows = strip_rows(ny_total, comm.world_size)
        if strip.shape[0] != rows:       # pad a short last strip on the host
            padded = np.full((rows, strip.shape[1]), pad_value, dtype=strip.dtype)
            padded[:strip.shape[0]] = strip.get()
            strip = DeviceArray.from_numpy(padded, strip.device)
        return comm.allgather_rows(strip, ny_total).get()[:ny_total]
    return comm.allgather_rows(strip.get() if isinstance(strip, DeviceArray) else strip, ny_total)


_KEYS = {0: "m0", 1: "m1", 2: "m2"}


def sharded_moments(strip_cube, ny_total, comm, orders=(0, 1, 2)):
    """moment maps of a cube sharded by row strips: every rank computes the
    maps of ITS strip with the fused HIP kernel, then one all-gather stitches
    them.  Returns {order: (ny_total, nx) float64 ndarray} on every rank."""
    want = tuple(_KEYS[o] for o in orders)
    r = strip_cube._moment_device(want)
    return {o: _stitch(r[_KEYS[o]], ny_total, comm) for o in orders}


def sharded_spectral_smooth_moment(strip_cube, kernel, ny_total, comm, orders=(1,)):
    """config C3 (spectral_smooth -> moment) on row strips: the stencil runs along z, so strips need
    no halo and no exchange; every rank runs the FUSED smooth->moment kernels on its strip (the
    smoothed strip is never written, dask_spectral_cube.py:880-917 is lazy and keeps the mask) and
    one all-gather stitches the maps.  Returns {order: (ny_total, nx) float64 ndarray}."""
    sm = strip_cube.spectral_smooth(kernel)
    want = tuple(_KEYS[o] for o in orders)
    r = sm._moment_device(want, sm._fusable())
    return {o: _stitch(r[_KEYS[o]], ny_total, comm) for o in orders}


def sharded_spectral_interpolate(strip_cube, spectral_grid, **kw):
    """config C5, first half: spectral_interpolate is per spaxel (dask_spectral_cube.py:1250-1373), so
    the rank's strip is resampled in place - no halo, no exchange.  Returns the rank's new strip cube."""
    return strip_cube.spectral_interpolate(spectral_grid, **kw)


def _strip_pixel_map(wcs_in, header_out, rank, world_size, device):
    """the rank's rows [y0, y1) of the full target pixel map (device, float64) + their row range"""
    from . import ops
    from .wcs import SimpleWCS
    newwcs = header_out if isinstance(header_out, SimpleWCS) else SimpleWCS(header_out)
    hdr = newwcs.header
    ny_out, nx_out = int(hdr["NAXIS2"]), int(hdr["NAXIS1"])
    y0, y1 = strip_bounds(ny_out, world_size, rank)
    xs, ys = ops.wcs_pixel_map(wcs_in, newwcs, (ny_out, nx_out), device)   # 0.3 ms per 1024^2: every rank forms the whole map
    cut = lambda a: DeviceArray((y1 - y0, nx_out), np.float64, device, ptr=a.ptr + y0 * nx_out * 8, owner=a)  # noqa: E731
    return cut(xs), cut(ys), (y0, y1), (ny_out, nx_out), newwcs


def reproject_source_rows(wcs_in, shape_in, header_out, rank, world_size, device=0):
    """config C5, second half: rows [r0, r1) of the SOURCE image that the rank's strip of output rows
    reads (bilinear: the two rows around every in-image source coordinate, one row of margin so that a
    strip edge is never mistaken for the image border).  (0, 0) when the strip sees nothing.  The rank
    then loads only those rows (io_fits.load_cube(rows=...) / DeviceArray.rows)."""
    ny_in, nx_in = int(shape_in[-2]), int(shape_in[-1])
    xs, ys, _, _, _ = _strip_pixel_map(wcs_in, header_out, rank, world_size, device)
    if ys.shape[0] == 0:
        return 0, 0
    hx, hy = xs.get(), ys.get()
    inside = (hx >= -0.5) & (hx <= nx_in - 0.5) & (hy >= -0.5) & (hy <= ny_in - 0.5)
    if not inside.any():
        return 0, 0
    lo, hi = float(hy[inside].min()), float(hy[inside].max())
    return max(0, int(np.floor(lo)) - 1), min(ny_in, int(np.ceil(hi)) + 2)


def sharded_reproject(src_rows, r0, wcs_in, shape_in, header_out, rank, world_size, mask=None, fill=np.nan, lerp=None):
    """Reproject the rank's strip of OUTPUT rows (spectral_cube.py:2649-2746 sharded by output rows).

    src_rows: (nz, r1 - r0, nx_in) float32 DeviceArray = rows [r0, r1) of the source cube as given by
    reproject_source_rows (mask: MaskSpec over the same rows).  The strip's pixel map is the rank's
    rows of the full map with r0 subtracted from the row coordinate (exact in float64), so the result
    is bit-identical to the same rows of the unsharded reprojection.  Returns (out, footprint,
    (y0, y1)): (nz, y1 - y0, nx_out) float32 + (y1 - y0, nx_out) uint8 DeviceArrays; ranks whose
    strip sees no source pixel get NaN rows and a zero footprint.  *lerp*: (lo, t, inv_dx) of ops.lerp_plan - the
    spectral interpolation of config C5's first half folded into the same pass (ops.resample_bilinear_lerp: the rank
    reads its source rows once and writes its len(lo) output channels once)."""
    from . import ops
    device = src_rows.device
    xs, ys, (y0, y1), (ny_out, nx_out), _ = _strip_pixel_map(wcs_in, header_out, rank, world_size, device)
    nz = src_rows.shape[0] if lerp is None else len(lerp[0])
    if y1 == y0:
        return DeviceArray((nz, 0, nx_out), np.float32, device), DeviceArray((0, nx_out), np.uint8, device), (y0, y1)
    ys_local = DeviceArray.from_numpy(ys.get() - float(r0), device)        # a few MB; exact subtraction
    if src_rows.shape[1] == 0:
        out = DeviceArray.from_numpy(np.full((nz, y1 - y0, nx_out), np.nan, np.float32), device)
        return out, DeviceArray.zeros((y1 - y0, nx_out), np.uint8, device), (y0, y1)
    if lerp is not None:
        out, foot = ops.resample_bilinear_lerp(src_rows, xs, ys_local, lerp[0], lerp[1], lerp[2], fill=fill, mask=mask)
    else:
        out, foot = ops.resample_bilinear(src_rows, xs, ys_local, fill=fill, mask=mask)
    return out, foot, (y0, y1)


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


def sharded_statistics(strip_stats, transport):
    """statistics() of a row-sharded cube: every rank passes the record of ITS strip (ops.stats_global on
    the device, one pass), five numbers per rank travel through the rendezvous transport."""
    rec = {k: float(strip_stats[k]) for k in ("npts", "min", "max", "sum", "sumsq")}
    return combine_statistics(transport.allgather_object(rec))


def walk_key_histograms(q, count_pass, next_pass):
    """The host side of a whole-cube order statistic (numpy 'linear' percentile) over 32-bit order-preserving keys:
    four byte passes pin the key of rank floor(pos) down, a fifth gives the next larger key when the two order
    statistics differ.  *count_pass(prefix, pmask, shift)* returns the 256 counters of one pass summed over ALL
    ranks, *next_pass(prefix)* the smallest key above prefix over all ranks.  Returns (key_lo, key_hi, frac) or None
    when no sample is included.  (The same walk as spc_percentile_global_f32 does for one GPU.)"""
    prefix = pmask = 0
    n = below = eq = 0
    k = khi = 0
    frac = 0.0
    for p in range(4):
        shift = 24 - 8 * p
        h = [int(v) for v in count_pass(prefix, pmask, shift)]
        if p == 0:
            n = sum(h)
            if n == 0:
                return None
            pos = q / 100.0 * (n - 1)
            k = int(np.floor(pos))
            khi = min(int(np.ceil(pos)), n - 1)
            frac = pos - np.floor(pos)
        d = 0
        while d < 255 and k >= h[d]:
            k -= h[d]
            below += h[d]
            d += 1
        eq = h[d]
        prefix |= d << shift
        pmask |= 0xff << shift
    key_hi = prefix
    if khi >= below + eq:
        key_hi = int(next_pass(prefix))
    return prefix, key_hi, float(frac)


def sharded_percentile(strip, q, transport, mask=None, center=None, passes=None):
    """median / percentile / (with *center*) the MAD of a cube whose row strips live on different ranks
    (np.nanpercentile(cube, q) over everything; dask_spectral_cube.py:657-693 with the chunks on GPUs): every rank
    histograms ITS strip (ops.key_histogram, one streaming pass per key byte), the 256 counters of all ranks are added
    through the rendezvous transport (2 KB per rank and pass) and every rank walks the same totals.  *strip* is this
    rank's DeviceArray (nz, rows, nx) - rows may be 0.  Returns the same python float on every rank.
    *passes* = (key_histogram, key_next, key_to_float) replaces the device passes (the CPU tests of the exchange)."""
    from . import ops
    hist_fn, next_fn, unkey = passes or (ops.key_histogram, ops.key_next, ops.key_to_float)
    empty = strip.shape[0] == 0 or strip.shape[1] == 0 or strip.shape[2] == 0

    def count_pass(prefix, pmask, shift):
        mine = np.zeros(256, np.uint64) if empty else hist_fn(strip, prefix, pmask, shift, mask=mask, center=center)
        parts = transport.allgather_bytes(mine.tobytes())
        return np.sum([np.frombuffer(b, np.uint64) for b in parts], axis=0, dtype=np.uint64)

    def next_pass(prefix):
        mine = 0xffffffff if empty else next_fn(strip, prefix, mask=mask, center=center)
        return min(int.from_bytes(b, "little") for b in transport.allgather_bytes(int(mine).to_bytes(4, "little")))

    r = walk_key_histograms(float(q), count_pass, next_pass)
    if r is None:
        return float("nan")
    a, b, frac = unkey(r[0]), unkey(r[1]), r[2]
    with np.errstate(invalid="ignore"):
        return float(a if (float(q) == 50.0 and a == b) else (0.5 * (a + b) if frac == 0.5 else a + (b - a) * frac))


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
    full = _stitch(s0, ny_total, comm, pad_value=0.0)
    if not np.all(np.isfinite(full)):
        return None
    karr = kernel_array(kernel, 2)
    c0 = ops.map_conv2d(DeviceArray.from_numpy(np.ascontiguousarray(full), strip_cube.device), karr).get()
    return strip_cube._pix_size_slice(0) * c0
