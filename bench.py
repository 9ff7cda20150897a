#!/usr/bin/env python
"""bench.py - headline benchmark: fused moment0+moment1+moment2 of a masked
fp32 cube (BASELINE.json configs[1]: 1024x1024x1024, boolean uint8 mask).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (ONE launch of the fused HIP moment
kernel -> three float64 maps) over the rank's device-resident cube; with N>1
every rank owns a (nz, ny, nx) row strip of a cube N times taller (weak
scaling) and the step ends with ONE RCCL all-gather of the three map strips.
Prints ONE JSON line (rank 0) with the metric, the roofline of the dominant
kernel measured live with HIP events on the kernel's stream, and a CPU
baseline (numpy restatement of the reference's arithmetic = oracle, threads
over spaxel chunks like dask's `threads` scheduler) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--shape", type=int, nargs=3, default=[1024, 1024, 1024], metavar=("NZ", "NY", "NX"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU baseline time budget")
    ap.add_argument("--verify", action="store_true", default=True)
    return ap.parse_args()


def fill_cube_on_device(cube, mask, shape, seed, y_offset):
    """Seeded synthetic strip generated row-block by row-block on the host
    (thread pool; every block has its own seed) and staged to HBM - never
    timed.  Data: Gaussian line per spaxel + noise (spectral_cube_amd.synth,
    SURVEY.md section 8d); mask: data > 2*noise with a 1 % flip, one fully
    masked 8x8 block and one NaN-input block."""
    import ctypes as C
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
        return (blk, m) if y0 == 0 else (None, float(m.mean()))

    keep = []
    with ThreadPoolExecutor(min(16, len(os.sched_getaffinity(0)))) as ex:
        res = list(ex.map(one, range(0, ny, rows)))
    _lib.call("spc_device_sync", cube.device)
    del keep[:]
    return res[0]


def cpu_baseline(shape, seconds):
    """numpy float64 restatement of the reference's Dask arithmetic (the
    oracle), chunked over spaxels with a thread pool = what
    `use_dask_scheduler('threads')` does.  Bounded sample of the same
    workload: as many (nz, rows, nx) strips as fit in ~`seconds`."""
    import numpy as np
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

    dist = None
    torch = None
    # launched by torch.distributed.run (even with one rank): take the distributed path,
    # so that a 1-GPU box can exercise rendezvous + RCCL init + the all-gather stitch
    distributed = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ
                                and os.environ.get("SPC_BENCH_FORCE_DIST", "0") == "1")
    if distributed:
        # torch first: its bundled HIP/RCCL runtime is then the single runtime of the
        # process (libspcube_hip.so binds to the already loaded sonames)
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", init_method="env://")

    import numpy as np
    from spectral_cube_amd import _lib, ops, synth
    from spectral_cube_amd.device import DeviceArray, Event, Stream, synchronize
    from spectral_cube_amd.distributed import RcclComm, HostGatherComm, torch_bcast_bytes

    _lib.require_gpu()
    device = local_rank % _lib.device_count()
    nz, ny, nx = args.shape
    shape = (nz, ny, nx)
    vox_rank = nz * ny * nx

    cube = DeviceArray(shape, np.float32, device)
    maskd = DeviceArray(shape, np.uint8, device)
    first_blk = fill_cube_on_device(cube, maskd, shape, synth.SEEDS["C2"], rank * ny)
    v = synth.spectral_axis(nz)
    cen = v - v[0]
    cref = cen[nz // 2]
    d_cen = DeviceArray.from_numpy(cen - cref, device)
    mask = ops.MaskSpec(_lib.MASK_ARRAY, array=maskd)
    stream = Stream(device)
    out = {k: DeviceArray((ny, nx), np.float64, device) for k in ("m0", "m1", "m2")}
    need = _lib.load().spc_moments_workspace_bytes(nz, ny, nx)
    ws = DeviceArray((max(need, 1),), np.uint8, device)

    comm, stitch = None, "none"
    recv = None
    if distributed:
        try:
            comm = RcclComm(device, rank, world, torch_bcast_bytes())
            stitch = "rccl"
        except Exception as exc:          # loud, reported fallback for the STITCH only
            print("[bench] rank %d: RCCL init failed (%s); stitching through gloo on the host" % (rank, exc),
                  file=sys.stderr, flush=True)
            comm = HostGatherComm()
            stitch = "gloo-host-fallback"
        flags = [None] * world
        dist.all_gather_object(flags, stitch)
        if any(f != "rccl" for f in flags):
            if stitch == "rccl":
                comm.close()
                comm = HostGatherComm()
            stitch = "gloo-host-fallback"
        # Two send buffers (three map strips each) + two receive buffers: the stitch of step k
        # runs on its own stream while the kernel of step k+1 fills the other buffer - the
        # all-gather (3 x 8 MiB per rank and step) is latency/link bound and would otherwise
        # serialise behind every 0.9 ms kernel.
        sends = [DeviceArray((3, ny, nx), np.float64, device) for _ in range(2)]
        outs = []
        for snd in sends:
            outs.append({k: DeviceArray((ny, nx), np.float64, device, ptr=snd.ptr + i * ny * nx * 8, owner=snd)
                         for i, k in enumerate(("m0", "m1", "m2"))})
        recvs = [DeviceArray((world, 3, ny, nx), np.float64, device) for _ in range(2)]   # map k = recv[:, k] along y
        comm_stream = Stream(device)
        ev_kernel = [Event(device), Event(device)]     # kernel of buffer b finished
        ev_comm = [Event(device), Event(device)]       # stitch of buffer b finished (buffer reusable)
        out = outs[0]
    step_no = [0]

    def step():
        if not distributed:
            ops.moments(cube, d_cen, dv=500.0, m1_add=cref + v[0], mask=mask, want=("m0", "m1", "m2"),
                        stream=stream, workspace=ws, out=out)
            return
        b = step_no[0] & 1
        if step_no[0] >= 2:
            stream.wait_event(ev_comm[b])               # the stitch that last used this buffer is done
        ops.moments(cube, d_cen, dv=500.0, m1_add=cref + v[0], mask=mask, want=("m0", "m1", "m2"),
                    stream=stream, workspace=ws, out=outs[b])
        ev_kernel[b].record(stream)
        if stitch == "rccl":
            comm_stream.wait_event(ev_kernel[b])
            comm.allgather_rows_device(sends[b], recvs[b], comm_stream)
            ev_comm[b].record(comm_stream)
        else:
            stream.synchronize()
            comm.allgather_rows(sends[b].get().reshape(3 * ny, nx), 3 * ny * world)
            ev_comm[b].record(stream)
        step_no[0] += 1

    def barrier():
        stream.synchronize()
        if distributed and stitch == "rccl":
            comm_stream.synchronize()
        synchronize(device)
        if torch is not None and torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

    # ---- dominant kernel timed alone with HIP events on ITS stream ------------------
    e0, e1 = Event(device), Event(device)
    kt = []
    for _ in range(min(20, max(5, args.steps))):
        e0.record(stream)
        ops.moments(cube, d_cen, dv=500.0, m1_add=cref + v[0], mask=mask, want=("m0", "m1", "m2"),
                    stream=stream, workspace=ws, out=out)
        e1.record(stream)
        e1.synchronize()
        kt.append(e0.elapsed_ms(e1))
    kt.sort()
    k_ms = float(np.mean(kt))
    alg_bytes = vox_rank * 5 + ny * nx * 24          # 4 B data + 1 B mask per voxel, 3 fp64 maps out
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC passes committed under profiles/ (separate
    # rocprofv3 --pmc runs of this same command; FETCH_SIZE x2 gfx950 correction)
    traffic, traffic_src = None, None
    pmc_file = os.path.join(REPO, "profiles", "r01_moments_c2_pmc.json")
    if os.path.exists(pmc_file) and shape == (1024, 1024, 1024):
        with open(pmc_file) as fh:
            traffic = json.load(fh)["hbm_traffic_bytes_per_launch"]
        traffic_src = "profiles/r01_moments_c2_pmc.json"

    # ---- verification of the timed outputs (first rows vs the oracle; never timed) ----
    verify = None
    if rank == 0:
        sys.path.insert(0, os.path.join(REPO, "oracle"))
        import oracle_np as O
        blk, m = first_blk
        rows = blk.shape[1]
        e = O.moments012(blk, m.astype(bool), cen, 500.0, v[0])
        got = [out[k].get()[:rows] for k in ("m0", "m1", "m2")]
        errs = []
        with np.errstate(all="ignore"):
            for g_, e_, sc in zip(got, e, (np.nanmax(np.abs(e[0])), 500.0 * nz, np.nanmax(np.abs(e[2])))):
                assert np.array_equal(np.isnan(g_), np.isnan(e_)), "NaN pattern mismatch vs oracle"
                ok = np.isfinite(e_)
                errs.append(float(np.abs(g_[ok] - e_[ok]).max() / sc))
        assert max(errs) <= 1e-5, errs
        verify = {"rows_checked": int(rows), "max_scaled_err_m0_m1_m2": errs, "nan_pattern": "identical"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(shape, args.cpu_seconds)

    if rank == 0:
        total_vox = vox_rank * world * args.steps
        value = total_vox / elapsed / 1e6
        info = _lib  # noqa
        from spectral_cube_amd.device import device_info
        line = {
            "metric": "Mvoxel/s (and % HBM roofline) for moment0/1/2 on masked fp32 cube, 1/2/4/8 GPU",
            "value": value, "unit": "Mvoxel/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: %dx%dx%d fp32 cube per GPU, uint8 boolean mask, fused "
                                   "moment0+moment1+moment2 (one kernel launch, three float64 maps)" % shape,
                       "input_dtype": "f32 cube + u8 mask, sums carried in f64, f64 maps out",
                       "mask_valid_fraction": float(first_blk[1].mean()), "stitch": stitch,
                       "stitch_overlap": "all-gather of step k overlaps the kernel of step k+1 (double buffered)" if distributed else None,
                       "sharding": "row strips of a %dx%dx%d cube" % (nz, ny * world, nx) if world > 1 else "none",
                       "device": device_info(device)["name"] or device_info(device)["arch"]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic_src,
                         "kernel": "moments_kernel<VEC=4,ZW=4,U=8,ARR,noEXT,NT>",
                         "kernel_ms": k_ms, "algorithmic_bytes": alg_bytes},
            "cpu_baseline": cpu,
            "verify": verify,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
