#!/usr/bin/env python
"""bench.py - headline benchmark: fused moment0+moment1+moment2 of a masked fp32 cube.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (ONE launch of the fused HIP moment kernel -> three float64
maps) over the rank's device-resident cube.

N = 1   headline `value`: BASELINE.json configs[1] (1024x1024x1024 fp32 + uint8 mask).  The same
        line carries a `north_star` record: the 4096x2048x2048 fp32 + uint8 cube of the north star
        (80 GiB, resident on the one GPU), kernel time by HIP events, Mvoxel/s and fraction of the
        8 TB/s roofline, checked against the oracle.
N > 1   STRONG scaling of that fixed 4096x2048x2048 cube: rank r owns rows [r*2048/N, (r+1)*2048/N)
        (x contiguity kept), every step = the rank's kernel followed, on the same stream, by ONE
        RCCL all-gather that stitches the three maps on every rank - the latency of one moment()
        call, nothing overlapped.  `value` = cube voxels * steps / max-over-ranks time.  The line
        also reports the pipelined rate (all-gather of step k under the kernel of step k+1, what a
        stream of cubes gets) and the kernel / all-gather times alone.

Rank 0 prints ONE JSON line: metric, roofline of the dominant kernel measured live with HIP events
on the kernel's stream, and (N = 1) a CPU baseline: the numpy restatement of the reference's
arithmetic (oracle), threads over spaxel chunks like dask's `threads` scheduler, bounded sample.
No torch: the launcher's RANK / WORLD_SIZE / LOCAL_RANK are read from the environment and the
ranks meet through spectral_cube_amd.rendezvous (files in a per-launch directory).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

METRIC = "Mvoxel/s (and % HBM roofline) for moment0/1/2 on masked fp32 cube, 1/2/4/8 GPU"
NORTH_STAR = (4096, 2048, 2048)
PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--shape", type=int, nargs=3, default=[1024, 1024, 1024], metavar=("NZ", "NY", "NX"),
                    help="headline cube at N=1 (configs[1])")
    ap.add_argument("--north-star-shape", type=int, nargs=3, default=list(NORTH_STAR), metavar=("NZ", "NY", "NX"))
    ap.add_argument("--no-north-star", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU baseline time budget")
    return ap.parse_args()


# ---- synthetic inputs (never timed) ------------------------------------------------------------
def fill_cube_on_device(cube, mask, shape, seed, y_offset):
    """Seeded synthetic strip generated row-block by row-block on the host (thread pool; every
    block has its own seed) and staged to HBM.  Data: Gaussian line per spaxel + noise
    (spectral_cube_amd.synth); mask: data > 2*noise with a 1 % flip, one
    fully masked 8x8 block and one NaN-input block.  Returns (first block, its mask, valid
    fraction of the WHOLE cube)."""
    from concurrent.futures import ThreadPoolExecutor
    import numpy as np
    from spectral_cube_amd import _lib, synth
    nz, ny, nx = shape
    rows = 16

    def one(y0):
        y1 = min(ny, y0 + rows)
        blk = synth.gaussian_line_cube((nz, y1 - y0, nx), seed + 1000 * (y_offset + y0), chunk_rows=rows)
        m = synth.boolean_mask(blk, seed + 1000 * (y_offset + y0))
        if not (y_offset == 0 and y0 == 0):
            # boolean_mask() blanks an 8x8 block in every strip; keep only the first one
            m[:, :8, :8] = (blk[:, :8, :8] > 1.0).view(np.uint8)
        elif y1 - y0 >= 16 and nx >= 16:
            blk[:, 8:16, 8:16] = np.nan                     # NaN-input block
        # strided H2D: rows y0:y1 of every plane
        _lib.call("spc_memcpy3d_h2d", cube.device, C.c_void_p(cube.ptr + y0 * nx * 4), nx * 4, ny * nx * 4,
                  blk.ctypes.data_as(C.c_void_p), nx * 4, (y1 - y0) * nx * 4, nx * 4, y1 - y0, nz, None)
        _lib.call("spc_memcpy3d_h2d", mask.device, C.c_void_p(mask.ptr + y0 * nx), nx, ny * nx,
                  m.ctypes.data_as(C.c_void_p), nx, (y1 - y0) * nx, nx, y1 - y0, nz, None)
        keep.append((blk, m))       # host pages stay mapped until every staged copy has certainly drained (below)
        included = int(np.count_nonzero(m))
        return (blk, m, included) if y0 == 0 else (None, None, included)

    keep = []
    with ThreadPoolExecutor(min(16, len(os.sched_getaffinity(0)))) as ex:
        res = list(ex.map(one, range(0, ny, rows)))
    _lib.call("spc_device_sync", cube.device)
    del keep[:]
    return res[0][0], res[0][1], sum(r[2] for r in res) / float(nz * ny * nx)


def tiled_strip_on_device(shape, seed, device, tile_rows=16):
    """(nz, ny, nx) float32 cube + uint8 mask in HBM made of ONE seeded host tile (nz, tile_rows, nx)
    repeated along y by device-to-device copies (the 64 GiB north-star cube cannot be staged from the
    host within a benchmark's minutes; what a streaming kernel does per voxel does not depend on
    the repetition).  Returns (cube, mask, host tile, host tile mask)."""
    import numpy as np
    from spectral_cube_amd import _lib, synth
    from spectral_cube_amd.device import DeviceArray
    nz, ny, nx = shape
    tr = min(tile_rows, ny)
    tile = synth.gaussian_line_cube((nz, tr, nx), seed, chunk_rows=tr)
    tmask = synth.boolean_mask(tile, seed)
    if tr >= 16 and nx >= 16:
        tile[:, 8:16, 8:16] = np.nan
    cube, mask = DeviceArray(shape, np.float32, device), DeviceArray(shape, np.uint8, device)
    for dev, host, isz in ((cube, tile, 4), (mask, tmask, 1)):
        row = nx * isz
        _lib.call("spc_memcpy3d_h2d", device, C.c_void_p(dev.ptr), row, ny * row, host.ctypes.data_as(C.c_void_p),
                  row, tr * row, row, tr, nz, None)
        have = tr
        while have < ny:                                    # doubling along y
            n = min(have, ny - have)
            _lib.call("spc_memcpy3d_d2d", device, C.c_void_p(dev.ptr + have * row), row, ny * row,
                      C.c_void_p(dev.ptr), row, ny * row, row, n, nz, None)
            have += n
    _lib.call("spc_device_sync", device)
    return cube, mask, tile, tmask


def cpu_baseline(shape, seconds):
    """numpy float64 restatement of the reference's Dask arithmetic (the oracle), chunked over
    spaxels with a thread pool = what `use_dask_scheduler('threads')` does.  Bounded sample of the
    same workload: as many (nz, rows, nx) strips as fit in ~`seconds`."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import oracle_np as O
    from spectral_cube_amd import synth
    nz, ny, nx = shape
    cores = len(os.sched_getaffinity(0))
    rows = 4
    cen = synth.spectral_axis(nz)
    cen = cen - cen[0]

    def make(i):
        blk = synth.gaussian_line_cube((nz, rows, nx), 999 + i, chunk_rows=rows)
        return blk, synth.boolean_mask(blk, 999 + i).astype(bool)

    def work(item):
        blk, inc = item
        return [O.moment(blk, inc, o, cen, 500.0, world0=-1.0) for o in (0, 1, 2)]

    cores = min(cores, 64)                                  # numpy stops scaling long before 256 threads
    items = [make(i) for i in range(cores)]
    t0 = time.perf_counter()
    work(items[0])
    t1 = time.perf_counter() - t0                           # single-thread time per strip
    rounds = 0
    with ThreadPoolExecutor(cores) as ex:
        t0 = time.perf_counter()
        while True:
            list(ex.map(work, items))                       # every core gets one strip per round
            rounds += 1
            dt = time.perf_counter() - t0
            if dt >= seconds or rounds >= 64:
                break
    vox = rounds * cores * nz * rows * nx
    return {"value": vox / dt / 1e6, "unit": "Mvoxel/s", "cores": cores, "kind": "port",
            "single_thread_value": nz * rows * nx / t1 / 1e6,
            "sample": "%d strips of %dx%dx%d voxels (moment 0,1,2 = three reference passes each), "
                      "numpy float64 oracle, ThreadPool(%d)" % (rounds * cores, nz, rows, nx, cores),
            # the reference itself only runs in the build container; measured there next to this port
            "reference_anchor": "profiles/r01_reference_cpu_buildbox.txt (8 cores, 256^3, moment 0+1+2): Dask class "
                                "4.6 Mvoxel/s, NumPy class 6.8, this port on one thread 8.7"}


# ---- helpers ---------------------------------------------------------------------------------------
class Workload:
    """one rank's resident cube + everything a step needs"""

    def __init__(self, cube, maskd, device):
        import numpy as np
        from spectral_cube_amd import _lib, ops, synth
        from spectral_cube_amd.device import DeviceArray, Stream
        self.np, self.ops = np, ops
        self.cube, self.maskd, self.device = cube, maskd, device
        nz, ny, nx = cube.shape
        self.shape = (nz, ny, nx)
        self.v = synth.spectral_axis(nz)
        self.cen = self.v - self.v[0]
        self.cref = self.cen[nz // 2]
        self.d_cen = DeviceArray.from_numpy(self.cen - self.cref, device)
        self.mask = ops.MaskSpec(_lib.MASK_ARRAY, array=maskd)
        self.stream = Stream(device)
        need = _lib.load().spc_moments_workspace_bytes(nz, ny, nx)
        self.ws = DeviceArray((max(need, 1),), np.uint8, device)
        self.alg_bytes = nz * ny * nx * 5 + ny * nx * 24     # 4 B data + 1 B mask per voxel, 3 fp64 maps out

    def outputs(self, send=None):
        """three (ny, nx) float64 maps; laid out back to back in `send` ((3, ny, nx)) when given"""
        from spectral_cube_amd.device import DeviceArray
        np = self.np
        nz, ny, nx = self.shape
        if send is None:
            return {k: DeviceArray((ny, nx), np.float64, self.device) for k in ("m0", "m1", "m2")}
        return {k: DeviceArray((ny, nx), np.float64, self.device, ptr=send.ptr + i * ny * nx * 8, owner=send)
                for i, k in enumerate(("m0", "m1", "m2"))}

    def launch(self, out, stream=None):
        self.ops.moments(self.cube, self.d_cen, dv=500.0, m1_add=self.cref + self.v[0], mask=self.mask,
                         want=("m0", "m1", "m2"), stream=stream or self.stream, workspace=self.ws, out=out)

    def kernel_ms(self, out, n):
        """mean launch duration by HIP events on the kernel's stream"""
        from spectral_cube_amd.device import Event
        e0, e1 = Event(self.device), Event(self.device)
        kt = []
        for _ in range(n):
            e0.record(self.stream)
            self.launch(out)
            e1.record(self.stream)
            e1.synchronize()
            kt.append(e0.elapsed_ms(e1))
        return float(self.np.mean(kt))

    def verify(self, maps, tile, tmask, rows):
        """first `rows` rows of (m0, m1, m2) host maps vs the oracle on the host tile"""
        np = self.np
        sys.path.insert(0, os.path.join(REPO, "oracle"))
        import oracle_np as O
        nz = self.shape[0]
        e = O.moments012(tile[:, :rows], tmask[:, :rows].astype(bool), self.cen, 500.0, self.v[0])
        errs = []
        with np.errstate(all="ignore"):
            for g_, e_, sc in zip(maps, e, (np.nanmax(np.abs(e[0])), 500.0 * nz, np.nanmax(np.abs(e[2])))):
                g_ = g_[:rows]
                assert np.array_equal(np.isnan(g_), np.isnan(e_)), "NaN pattern mismatch vs oracle"
                ok = np.isfinite(e_)
                errs.append(float(np.abs(g_[ok] - e_[ok]).max() / sc))
        assert max(errs) <= 1e-5, errs
        return {"rows_checked": int(rows), "max_scaled_err_m0_m1_m2": errs, "nan_pattern": "identical"}


def pmc_traffic(shape):
    """HBM bytes per launch from the PMC passes committed under profiles/ (separate rocprofv3 --pmc
    runs of this same command; FETCH_SIZE x2 gfx950 correction)"""
    for name in ("r02_moments_c2_pmc.json", "r01_moments_c2_pmc.json"):
        f = os.path.join(REPO, "profiles", name)
        if os.path.exists(f) and tuple(shape) == (1024, 1024, 1024):
            with open(f) as fh:
                return json.load(fh)["hbm_traffic_bytes_per_launch"], "profiles/" + name
    return None, None


def roofline(wl, k_ms, traffic=None, traffic_src=None):
    achieved = wl.alg_bytes / (k_ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": achieved, "peak": PEAK_GBS, "unit": "GB/s", "frac": achieved / PEAK_GBS,
            "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
            "kernel": "moments_kernel<VEC=4,ZW=4,U=8,ARR,noEXT,NT>", "kernel_ms": k_ms,
            "algorithmic_bytes": wl.alg_bytes}


# ---- N = 1 -------------------------------------------------------------------------------------------
def run_single(args, device):
    import gc
    import numpy as np
    from spectral_cube_amd import synth
    from spectral_cube_amd.device import DeviceArray, device_info, pool_trim, synchronize
    shape = tuple(args.shape)
    nz, ny, nx = shape
    cube, maskd = DeviceArray(shape, np.float32, device), DeviceArray(shape, np.uint8, device)
    blk, m, valid_frac = fill_cube_on_device(cube, maskd, shape, synth.SEEDS["C2"], 0)
    wl = Workload(cube, maskd, device)
    out = wl.outputs()

    def barrier():
        wl.stream.synchronize()
        synchronize(device)

    for _ in range(args.warmup):
        wl.launch(out)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.launch(out)
    barrier()
    elapsed = time.perf_counter() - t0

    k_ms = wl.kernel_ms(out, min(20, max(5, args.steps)))
    traffic, traffic_src = pmc_traffic(shape)
    verify = wl.verify([out[k].get() for k in ("m0", "m1", "m2")], blk, m, blk.shape[1])
    line = {
        "metric": METRIC, "value": nz * ny * nx * args.steps / elapsed / 1e6, "unit": "Mvoxel/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "configs[1]: %dx%dx%d fp32 cube, uint8 boolean mask, fused moment0+moment1+moment2 "
                               "(one kernel launch, three float64 maps)" % shape,
                   "input_dtype": "f32 cube + u8 mask, sums carried in f64, f64 maps out",
                   "mask_valid_fraction": valid_frac, "stitch": "none", "sharding": "none",
                   "device": device_info(device)["name"] or device_info(device)["arch"]},
        "roofline": roofline(wl, k_ms, traffic, traffic_src),
        "verify": verify,
    }
    # ---- north-star record: the fixed 4096x2048x2048 cube on this one GPU ------------------------
    del wl, out, cube, maskd, blk, m
    gc.collect()
    pool_trim(device)
    ns_shape = tuple(args.north_star_shape)
    if not args.no_north_star:
        need = ns_shape[0] * ns_shape[1] * ns_shape[2] * 5 * 1.03
        free = device_info(device)["free_mem"]
        if free < need:
            line["north_star"] = {"skipped": "needs %.0f GiB of HBM, %.0f GiB free" % (need / 2**30, free / 2**30)}
        else:
            line["north_star"] = north_star_record(ns_shape, device)
    return line


def north_star_record(shape, device):
    import gc
    from spectral_cube_amd import synth
    from spectral_cube_amd.device import pool_trim
    import numpy as np
    cube, maskd, tile, tmask = tiled_strip_on_device(shape, synth.SEEDS["C4"], device)
    wl = Workload(cube, maskd, device)
    out = wl.outputs()
    for _ in range(2):
        wl.launch(out)
    wl.stream.synchronize()
    k_ms = wl.kernel_ms(out, 10)
    verify = wl.verify([out[k].get() for k in ("m0", "m1", "m2")], tile, tmask, 4)
    nz, ny, nx = shape
    rec = {"workload": "north star: %dx%dx%d fp32 cube + uint8 mask resident on ONE GPU, fused moment0+1+2, "
                       "device-tiled synthetic data (one seeded %d-row host tile repeated along y)" % (shape + (tile.shape[1],)),
           "kernel_ms": k_ms, "value": nz * ny * nx / (k_ms * 1e-3) / 1e6, "unit": "Mvoxel/s",
           "mask_valid_fraction": float(np.count_nonzero(tmask)) / tmask.size,
           "roofline": roofline(wl, k_ms), "verify": verify, "target_frac": 0.60}
    del wl, out, cube, maskd
    gc.collect()
    pool_trim(device)
    return rec


# ---- N > 1: strong scaling of the north-star cube -----------------------------------------------
def run_sharded(args, device, rdv):
    import numpy as np
    from spectral_cube_amd import synth
    from spectral_cube_amd.device import DeviceArray, Event, Stream, device_info, synchronize
    from spectral_cube_amd.distributed import HostGatherComm, RcclComm, strip_bounds
    rank, world = rdv.rank, rdv.world_size
    NZ, NY, NX = tuple(args.north_star_shape)
    if NY % world:
        raise SystemExit("the %d rows of the cube must divide over %d ranks" % (NY, world))
    y0, y1 = strip_bounds(NY, world, rank)
    rows = y1 - y0
    cube, maskd, tile, tmask = tiled_strip_on_device((NZ, rows, NX), synth.SEEDS["C4"] + 17 * rank, device)
    wl = Workload(cube, maskd, device)

    try:
        comm, stitch = RcclComm(device, rdv), "rccl"
    except Exception as exc:          # loud, reported fallback for the STITCH only
        print("[bench] rank %d: RCCL init failed (%s); stitching through the host rendezvous" % (rank, exc),
              file=sys.stderr, flush=True)
        comm, stitch = None, "host-fallback"
    flags = rdv.allgather_object(stitch)
    if any(f != "rccl" for f in flags):
        if comm is not None:
            comm.close()
        comm, stitch = HostGatherComm(rdv), "host-fallback"

    # send buffer = the rank's three map strips back to back; receive buffer (world, 3, rows, NX):
    # map k of the whole cube = recv[:, k] read along y
    sends = [DeviceArray((3, rows, NX), np.float64, device) for _ in range(2)]
    outs = [wl.outputs(s) for s in sends]
    recvs = [DeviceArray((world, 3, rows, NX), np.float64, device) for _ in range(2)]
    comm_stream = Stream(device)

    def gather(b, stream):
        if stitch == "rccl":
            comm.allgather_rows_device(sends[b], recvs[b], stream)
        else:
            stream.synchronize()
            full = comm.allgather_rows(sends[b].get().reshape(3 * rows, NX), 3 * rows * world)
            recvs[b].upload(full.reshape(world, 3, rows, NX))

    def barrier():
        wl.stream.synchronize()
        comm_stream.synchronize()
        synchronize(device)
        rdv.barrier()

    def timed(step_fn):
        for i in range(args.warmup):
            step_fn(i)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step_fn(args.warmup + i)
        barrier()
        return max(rdv.allgather_object(time.perf_counter() - t0))

    # (1) one call = kernel, then the stitch, on ONE stream: nothing overlaps (the contract's `value`)
    def step_serial(i):
        wl.launch(outs[0])
        gather(0, wl.stream)
    elapsed = timed(step_serial)

    # (1b) the same ONE call with the stitch hidden inside it (distributed.ChunkedMoments): the rank's rows in 4 blocks,
    # the all-gather of a block's three maps on a second stream under the kernel of the next block.  Nothing is carried
    # across calls: a call starts when the previous one has completed.  Needs the device all-gather.
    chunked, elapsed_chunked = None, None
    if stitch == "rccl" and rows % 64 == 0:
        from spectral_cube_amd.distributed import ChunkedMoments
        chunked = ChunkedMoments(cube, maskd, wl.d_cen, 500.0, wl.cref + wl.v[0], comm, chunks=4, workspace=wl.ws)

        def step_chunked(i):
            chunked(wl.stream, comm_stream)
        elapsed_chunked = timed(step_chunked)

    # (2) pipelined: the stitch of step k on its own stream under the kernel of step k+1
    ev_kernel, ev_comm = [Event(device), Event(device)], [Event(device), Event(device)]
    count = [0]

    def step_pipe(i):
        b = count[0] & 1
        if count[0] >= 2:
            wl.stream.wait_event(ev_comm[b])               # the stitch that last used this buffer is done
        wl.launch(outs[b])
        ev_kernel[b].record(wl.stream)
        if stitch == "rccl":
            comm_stream.wait_event(ev_kernel[b])
            gather(b, comm_stream)
            ev_comm[b].record(comm_stream)
        else:
            gather(b, wl.stream)
            ev_comm[b].record(wl.stream)
        count[0] += 1
    elapsed_pipe = timed(step_pipe)

    # (3) the two parts alone, HIP events on their stream
    k_ms = wl.kernel_ms(outs[0], min(20, max(5, args.steps)))
    e0, e1 = Event(device), Event(device)
    gt = []
    for _ in range(10):
        rdv.barrier()
        e0.record(wl.stream)
        gather(0, wl.stream)
        e1.record(wl.stream)
        e1.synchronize()
        gt.append(e0.elapsed_ms(e1))
    g_ms = max(rdv.allgather_object(float(np.mean(gt))))
    k_ms_max = max(rdv.allgather_object(k_ms))

    # ---- every rank checks ITS rows of the stitched maps against the oracle on its own tile -------
    wl.launch(outs[0])
    gather(0, wl.stream)
    wl.stream.synchronize()
    full = recvs[0].get()                                   # (world, 3, rows, NX)
    verify = wl.verify([full[rank, k] for k in range(3)], tile, tmask, 4)
    if chunked is not None:                                 # and the maps of the chunked call: every block of this rank
        maps = chunked(wl.stream, comm_stream)
        wl.stream.synchronize()
        hm = [maps[k].get() for k in ("m0", "m1", "m2")]
        for c in range(chunked.chunks):
            g0, g1 = chunked.global_rows(rank, c)
            wl.verify([m[g0:g1] for m in hm], tile, tmask, 4)
            # block c of this rank = local rows [c * rc, (c + 1) * rc) = the strip's rows in the one-launch maps
            for k in range(3):
                assert np.array_equal(hm[k][g0:g1], full[rank, k][c * chunked.rc:(c + 1) * chunked.rc], equal_nan=True), \
                    "chunked call disagrees with the one-launch call"
    # and that every rank holds the SAME stitched maps
    digest = [float(np.nansum(full[:, k])) for k in range(3)]
    digests = rdv.allgather_object(digest)
    assert all(d == digests[0] for d in digests), "ranks disagree on the stitched maps"
    verifies = rdv.allgather_object(verify)
    vfrac = rdv.allgather_object(float(np.count_nonzero(tmask)) / tmask.size)
    if comm is not None and stitch == "rccl":
        comm.close()

    total = NZ * NY * NX
    # the contract's value: ONE call at a time; with the device all-gather the call hides its stitch under its own
    # kernel (chunked), otherwise kernel then stitch
    best = elapsed if elapsed_chunked is None else min(elapsed, elapsed_chunked)
    call_form = "rows in 4 blocks, all-gather of a block under the kernel of the next" if best != elapsed else "one launch, then one all-gather"
    line = {
        "metric": METRIC, "value": total * args.steps / best / 1e6, "unit": "Mvoxel/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": best / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "north star: fixed %dx%dx%d fp32 cube + uint8 mask sharded by row strips over %d GPUs "
                               "(%d rows each), fused moment0+1+2 per strip + ONE all-gather stitching the three "
                               "float64 maps on every rank; ONE call at a time%s" % (
                                   NZ, NY, NX, world, rows,
                                   " - the rank's rows in 4 blocks (block-cyclic ownership), the all-gather of a block's maps "
                                   "under the kernel of the next block, nothing carried across calls" if best != elapsed
                                   else ", kernel then stitch, nothing overlapped"),
                   "input_dtype": "f32 cube + u8 mask, sums carried in f64, f64 maps out",
                   "mask_valid_fraction": float(np.mean(vfrac)), "stitch": stitch,
                   "sharding": "row strips (nz, %d, nx) of a %dx%dx%d cube" % (rows, NZ, NY, NX),
                   "allgather_bytes_per_rank": 3 * rows * NX * 8,
                   "device": device_info(device)["name"] or device_info(device)["arch"]},
        "per_call": {"kernel_ms": k_ms_max, "allgather_ms": g_ms, "latency_ms": best / args.steps * 1e3, "form": call_form,
                     "latency_unoverlapped_ms": elapsed / args.steps * 1e3,
                     "latency_chunked_ms": None if elapsed_chunked is None else elapsed_chunked / args.steps * 1e3},
        "pipelined": {"value": total * args.steps / elapsed_pipe / 1e6, "unit": "Mvoxel/s",
                      "ms_per_step": elapsed_pipe / args.steps * 1e3,
                      "note": "all-gather of step k on its own stream under the kernel of step k+1 (double buffered)"},
        "roofline": roofline(wl, k_ms_max),
        "cpu_baseline": None,
        "verify": {"per_rank": verifies, "stitched_maps_identical_on_all_ranks": True},
    }
    return line


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                     "--master-addr 127.0.0.1 --master-port 29533 bench.py --gpus %d ..." % (args.gpus, args.gpus))
        args.gpus = world
    from spectral_cube_amd import _lib
    from spectral_cube_amd.rendezvous import FileRendezvous, SingleProcess
    _lib.require_gpu()
    device = local_rank % _lib.device_count()
    # SPC_BENCH_FORCE_DIST=1: take the sharded path with one rank (rendezvous + RCCL init + stitch on a 1-GPU box)
    sharded = world > 1 or os.environ.get("SPC_BENCH_FORCE_DIST", "0") == "1"
    if sharded:
        rdv = FileRendezvous.from_env() if world > 1 else SingleProcess()
        line = run_sharded(args, device, rdv)
        rdv.close()
    else:
        line = run_single(args, device)
        line["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(tuple(args.shape), args.cpu_seconds)
    # whatever the native libraries still hold in the C stdio buffer (RCCL prints a version banner at init) goes out first:
    # the JSON line is the last line of stdout
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
