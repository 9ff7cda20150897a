#!/usr/bin/env python
"""bench.py - headline benchmark: fused moment0+moment1+moment2 of a masked fp32 cube.

    python bench.py --gpus N --steps K --warmup W          (N > 1: bench.py starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (a launcher's ranks)

One "step" = one pass of the hot path (ONE launch of the fused HIP moment kernel -> three float64
maps) over the rank's device-resident cube.

The workload of `value` is the SAME at every N: the 4096x2048x2048 fp32 + uint8-mask cube the north
star states its target on (80 GiB: resident on one MI355X), so the 1 -> 8 curve divides like by like.
N = 1   K launches between barriers, wall clock -> `value`; the kernel alone by HIP events on its
        stream -> `roofline`; checked against the oracle.  Secondary scalars: BASELINE.json
        configs[1] (1024^3), and one compact row per configs[2..4] / SURVEY 8(f) record.
N > 1   STRONG scaling of that cube: rank r owns rows [r*2048/N, (r+1)*2048/N) (x contiguity
        kept), every step = the rank's kernel + the RCCL all-gather stitching the maps on every
        rank (one call at a time; the chunked form hides the stitch inside the call).

Rank 0 prints ONE compact JSON line (< 6 KB, asserted: the round-5 line had grown to 31 KB and the
driver could not parse it).  Everything else - every record with its statistics, strip terms,
oracle windows - goes to bench_records.json beside this script and, one line per record, to stderr.
N = 1 also times a CPU baseline: the numpy restatement of the reference's arithmetic (oracle),
threads over spaxel chunks like dask's `threads` scheduler, bounded sample of the same workload.
No torch: with WORLD_SIZE unset and --gpus N > 1 this script starts its N ranks itself (one process
per GPU, RANK / LOCAL_RANK / WORLD_SIZE in their environment, rank 0's JSON relayed as the last line
of stdout, non-zero exit status if any rank fails); under a launcher its RANK / WORLD_SIZE /
LOCAL_RANK are read from the environment.  The ranks meet through spectral_cube_amd.rendezvous
(files in a per-launch directory).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import threading
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

METRIC = "Mvoxel/s (and % HBM roofline) for moment0/1/2 on masked fp32 cube, 1/2/4/8 GPU"
NORTH_STAR = (4096, 2048, 2048)
PEAK_GBS = 8000.0


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--shape", "--north-star-shape", dest="shape", type=int, nargs=3, default=list(NORTH_STAR),
                    metavar=("NZ", "NY", "NX"), help="the headline cube (every N); default: the north-star cube")
    ap.add_argument("--configs1-shape", type=int, nargs=3, default=[1024, 1024, 1024], metavar=("NZ", "NY", "NX"),
                    help="BASELINE.json configs[1] (secondary record at N = 1)")
    ap.add_argument("--no-configs1", action="store_true", help="skip the configs[1] record and the SURVEY 8(f) rows on it")
    ap.add_argument("--no-strip-terms", action="store_true", help="skip the rank-strip shapes of the headline cube")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU baseline time budget")
    ap.add_argument("--no-configs", action="store_true", help="skip the configs[2..4] records")
    ap.add_argument("--configs-only", default=None, metavar="C3,C4,C5",
                    help="only the named config records (kernel work: before / after numbers)")
    ap.add_argument("--configs-scale", type=int, default=1, help="divide the config cubes' longest axis (smoke runs)")
    ap.add_argument("--records-file", default=os.path.join(REPO, "bench_records.json"))
    return ap.parse_args(argv)


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
    rows = max(1, min(4, (1 << 22) // (nz * nx)))           # strips of ~4 - 8 Mvoxel whatever the cube
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
    return {"value": sig(vox / dt / 1e6), "unit": "Mvoxel/s", "cores": cores, "kind": "port",
            "single_thread_value": sig(nz * rows * nx / t1 / 1e6), "seconds": sig(dt, 4),
            "sample": "%d strips of %dx%dx%d voxels of the headline cube (moment 0,1,2 = three reference passes each), "
                      "numpy float64 oracle, ThreadPool(%d)" % (rounds * cores, nz, rows, nx, cores),
            # the reference itself only runs in the build container; measured there next to this port
            "reference_anchor": "profiles/r01_reference_cpu_buildbox.txt (8 cores, 256^3): Dask class 4.6 Mvoxel/s, "
                                "NumPy class 6.8, this port on one thread 8.7"}


# ---- helpers ---------------------------------------------------------------------------------------
class Ms(float):
    """a duration in ms = the MEDIAN of the timed launches (median of >= 10 timed runs), carrying
    min / max / mean / n so that every `frac` of the line can be set beside the rocprofv3 trace under profiles/"""

    def __new__(cls, samples):
        import numpy as np
        a = np.asarray(list(samples), dtype=np.float64)
        self = super().__new__(cls, float(np.median(a)))
        self.stats = {"median": float(np.median(a)), "min": float(a.min()), "max": float(a.max()), "mean": float(a.mean()), "n": int(a.size)}
        return self


def ms_stats(ms):
    return getattr(ms, "stats", {"median": float(ms), "min": float(ms), "max": float(ms), "mean": float(ms), "n": 1})


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
        """median launch duration (Ms: + min / max / mean) by HIP events on the kernel's stream"""
        from spectral_cube_amd.device import Event
        e0, e1 = Event(self.device), Event(self.device)
        kt = []
        for _ in range(n):
            e0.record(self.stream)
            self.launch(out)
            e1.record(self.stream)
            e1.synchronize()
            kt.append(e0.elapsed_ms(e1))
        return Ms(kt)

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


PMC_FILE = os.path.join("profiles", "r06_pmc_traffic_by_record.json")
_PMC = None
PMC_ON = True         # (set False by a run at non-default shapes: the committed counters were taken at the default ones)
LINE_LIMIT = 6000     # bytes of the final stdout line (the driver keeps ~8 KB)

ARITH_MOMENTS = "f32 samples, f64 sums and maps (reference: float64)"
# what every timed kernel computes in (round-5 verdict, weak 1: the precision policy of DESIGN section 5, per record)
A_SPEC = "f32 samples x f64 taps, f64 accumulate, one rounding to f32 (astropy's float64 convolve, bit-level)"
A_SPEC_ALG = "f64 weights (taps folded into the moment coordinates), f64 sums"
A_SPAT_F32 = "f32 fma (v_pk_fma_f32), f32 accumulate (reference float64: 1e-5 contract, measured 2.5e-7)"
A_SPAT_SPLIT = "f16 hi+lo split of the f32 samples on v_mfma_f32_16x16x32_f16, f32 accumulate (1e-5 contract, measured 1e-6)"
A_FUSED012 = "f32 sums per chunk of <= 64 channels about the chunk's middle channel, shifted to the map's mean in f64"
A_LERP = "f32 difference, f64 blend, one rounding to f32 (numpy's promotion)"
A_BIL = "f64 pixel map, f32 weights and fma"
A_STATS = "f64 sums, exact count / extrema"
A_SELECT = "exact selection (comparisons only), numpy's f32 midpoint"
A_CLIP = "exact median, f64 mean / std, f32 bounds"


def library_sha256():
    """sha256 of the libspcube_hip.so this process loaded: PMC counters are only quoted for the library they were taken on"""
    import hashlib
    from spectral_cube_amd import _lib
    h = hashlib.sha256()
    with open(_lib.LIB_PATH, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


_LIB_SHA = None


def pmc_traffic(record):
    """(HBM bytes per launch, source) of the bench record with key `record` from the PMC passes committed under profiles/:
    separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of THIS command at THIS shape (tools/prof_bench.sh,
    summarised per (kernel, grid) = per record by tools/prof_bench_summary.py; FETCH_SIZE x 2: the gfx950 correction of
    MI355X_MICROARCH.md).  The file names the sha256 of the library the counters were taken on: (None, "stale: ...") when
    the loaded library differs, (None, None) for a record the file does not hold, or at other shapes than the default."""
    global _PMC, _LIB_SHA
    if not PMC_ON:
        return None, None
    if _PMC is None:
        try:
            with open(os.path.join(REPO, PMC_FILE)) as fh:
                _PMC = json.load(fh)
        except (OSError, ValueError):
            _PMC = {}
    rec = _PMC.get("records", {}).get(record)
    if not rec:
        return None, None
    if _LIB_SHA is None:
        try:
            _LIB_SHA = library_sha256()
        except Exception:                                   # (the CPU test of the line builder has no library)
            _LIB_SHA = "unknown"
    if _PMC.get("library_sha256") != _LIB_SHA:
        return None, "stale: %s was taken on library %s" % (PMC_FILE, str(_PMC.get("library_sha256"))[:12])
    return rec["hbm_traffic_bytes_per_launch"], "%s#%s" % (PMC_FILE, record)


def moments_zw(wl):
    """waves per block of the launch spc_moments_f32 picks (spc_moments.hip make_plan): 8 on planes up to 8 MiB with
    512 or more channels, else 4"""
    shape = getattr(wl, "shape", None)
    if not shape:
        return 4
    nz, ny, nx = shape
    return 8 if (nz >= 512 and ny * nx <= (1 << 21)) else 4


def roofline(wl, k_ms, traffic=None, traffic_src=None):
    achieved = wl.alg_bytes / (k_ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": achieved, "peak": PEAK_GBS, "unit": "GB/s", "frac": achieved / PEAK_GBS,
            "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
            "traffic_over_algorithmic": (traffic / wl.alg_bytes) if traffic else None,
            "kernel": "moments_kernel<VEC=4,ZW=%d,U=8,ARR,sums,PRED=0>" % moments_zw(wl), "kernel_ms": float(k_ms), "kernel_ms_stats": ms_stats(k_ms),
            "timing": "median of n launches, HIP events on the kernel's stream", "algorithmic_bytes": wl.alg_bytes,
            "arithmetic": ARITH_MOMENTS}


def sig(x, n=6):
    """x to n significant digits (the compact line); None and non-floats pass through"""
    if isinstance(x, float):
        return float("%.*g" % (n, x)) if x == x and abs(x) != float("inf") else None
    return x


def iter_records(detail):
    """every timed record of the run, in the order of the run"""
    for grp in ("next_rows", "C3", "C4", "C5", "wide"):
        g = detail.get(grp) if grp in ("next_rows", "wide") else (detail.get("configs") or {}).get(grp)
        recs = g if isinstance(g, list) else ([v for v in g.values() if isinstance(v, dict)] if isinstance(g, dict) else [])
        for r in recs:
            if isinstance(r, dict) and "kernel_ms" in r:
                yield r


def compact_line(detail):
    """The ONE stdout line from the full record: the contract's keys, `roofline` of the headline kernel, `cpu_baseline`,
    the configs[1] scalars and one [kernel_ms, frac, traffic / algorithmic] row per record under its short key.  Never
    larger than LINE_LIMIT bytes: the rows are dropped first, then the notes."""
    h = detail["headline"]
    rf = h["roofline"]
    st = rf.get("kernel_ms_stats") or {}
    roof = {"bound": rf["bound"], "achieved": sig(rf["achieved"]), "peak": rf["peak"], "unit": rf["unit"], "frac": sig(rf["frac"], 4),
            "traffic": rf.get("traffic"), "traffic_over_algorithmic": sig(rf.get("traffic_over_algorithmic"), 4),
            "traffic_source": rf.get("traffic_source"), "kernel": rf["kernel"], "kernel_ms": sig(rf["kernel_ms"]),
            "kernel_ms_min": sig(st.get("min")), "kernel_ms_max": sig(st.get("max")), "kernel_launches_timed": st.get("n"),
            "algorithmic_bytes": rf["algorithmic_bytes"], "arithmetic": rf.get("arithmetic"), "target_frac": 0.60,
            "timing": "HIP events on the kernel's stream, median"}
    if h.get("strip_terms"):        # the kernel term of the scaling model, measured on this GPU: [one launch, in blocks, blocks]
        roof["strip_kernel_ms_at_n"] = {str(t["n_gpus_modelled"]): [sig(t["kernel_ms"], 4), sig(t["blocks_ms"], 4), t["blocks"]]
                                        for t in h["strip_terms"]}
    c1 = detail.get("configs1")
    if isinstance(c1, dict) and "roofline" in c1:
        r1 = c1["roofline"]
        roof["configs1"] = {"workload": c1["workload"], "kernel_ms": sig(r1["kernel_ms"]), "frac": sig(r1["frac"], 4),
                            "mvoxel_per_s": sig(c1["value"]), "traffic_over_algorithmic": sig(r1.get("traffic_over_algorithmic"), 4),
                            "max_scaled_err": sig(max(c1["verify"]["max_scaled_err_m0_m1_m2"]), 3)}
    elif isinstance(c1, dict):
        roof["configs1"] = c1
    line = {"metric": METRIC, "value": sig(h["value"], 7), "unit": "Mvoxel/s", "n_gpus": detail["n_gpus"], "steps": detail["steps"],
            "warmup": detail["warmup"], "ms_per_step": sig(h["ms_per_step"]), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": h["config"], "roofline": roof,
            "cpu_baseline": detail.get("cpu_baseline"),
            "verify": h.get("verify_compact"), "records_file": detail.get("records_file"),
            "records_columns": ["kernel_ms", "frac_of_8TBps", "traffic_over_algorithmic"]}
    if detail.get("per_call"):
        line["per_call"] = {k: sig(v) for k, v in detail["per_call"].items()}
    rows = {}
    for r in iter_records(detail):
        rows[r.get("key") or r["name"][:24]] = [sig(r["kernel_ms"], 4), sig(r.get("frac"), 3), sig(r.get("traffic_over_algorithmic"), 3)]
    errs = [g for g in (detail.get("configs") or {}).items() if isinstance(g[1], dict) and "error" in g[1]]
    for g in ("next_rows", "wide"):
        if isinstance(detail.get(g), dict) and "error" in detail[g]:
            errs.append((g, detail[g]))
    if errs:
        line["errors"] = {k: str(v["error"])[:160] for k, v in errs}
    line["records"] = rows
    out = json.dumps(line)
    if len(out) > LINE_LIMIT:
        line["records"] = {"dropped": "line too long: see records_file"}
        out = json.dumps(line)
    if len(out) > LINE_LIMIT:
        for k in ("records_columns", "verify", "per_call", "errors"):
            line.pop(k, None)
        line["roofline"].pop("configs1", None)
        out = json.dumps(line)
    assert len(out) <= LINE_LIMIT, len(out)
    return line


def emit(detail, records_file):
    """full record -> records_file (+ one line per record on stderr); returns the compact line"""
    line = compact_line(detail)
    try:
        with open(records_file, "w") as fh:
            json.dump(dict(detail, line=line), fh, indent=1)
    except OSError as exc:
        print("[bench] could not write %s: %s" % (records_file, exc), file=sys.stderr, flush=True)
    for r in iter_records(detail):
        print("[bench] %-14s %9.3f ms  frac %5.3f  t/a %-5s  %s | %s" % (
            r.get("key", "-"), r["kernel_ms"], r.get("frac") or 0.0,
            "%.2f" % r["traffic_over_algorithmic"] if r.get("traffic_over_algorithmic") else "-", r["name"][:110],
            r.get("arithmetic", "")), file=sys.stderr, flush=True)
    return line


# ---- N = 1 -------------------------------------------------------------------------------------------
def fit_shape(shape, device):
    """the headline cube, or - on a device whose free HBM cannot hold it - the same planes with fewer channels"""
    from spectral_cube_amd.device import device_info
    nz, ny, nx = shape
    free = device_info(device)["free_mem"]
    while nz > 64 and nz * ny * nx * 5 * 1.03 > free:
        nz //= 2
    return (nz, ny, nx)


def run_single(args, device):
    import gc
    import numpy as np
    from spectral_cube_amd import synth
    from spectral_cube_amd.device import DeviceArray, device_info, pool_trim, synchronize
    want = tuple(args.shape)
    shape = fit_shape(want, device)
    detail = {"n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "records_file": os.path.basename(args.records_file)}
    # ---- headline: the north-star cube on this one GPU --------------------------------------------
    h = headline_record(shape, device, args)
    fitted = "" if shape == want else " (the %dx%dx%d cube does not fit this device's free HBM)" % want
    h["config"] = {"workload": "north star: %dx%dx%d fp32 cube + uint8 boolean mask resident on ONE GPU%s, fused moment0+moment1+moment2 "
                               "(one kernel launch, three float64 maps)" % (shape + (fitted,)),
                   "input_dtype": "f32 cube + u8 mask, sums carried in f64, f64 maps out", "arithmetic": ARITH_MOMENTS,
                   "mask_valid_fraction": sig(h["mask_valid_fraction"], 4), "stitch": "none", "sharding": "none",
                   "data_note": "device-tiled synthetic data: one seeded 16-row host tile repeated along y",
                   "device": device_info(device)["name"] or device_info(device)["arch"]}
    detail["headline"] = h
    # ---- BASELINE.json configs[1] + the SURVEY 8(f) rows on the same resident cube ---------------------
    if not args.no_configs1:
        c1shape = tuple(args.configs1_shape)
        nz, ny, nx = c1shape
        cube, maskd = DeviceArray(c1shape, np.float32, device), DeviceArray(c1shape, np.uint8, device)
        blk, m, valid_frac = fill_cube_on_device(cube, maskd, c1shape, synth.SEEDS["C2"], 0)
        wl = Workload(cube, maskd, device)
        out = wl.outputs()
        for _ in range(3):
            wl.launch(out)
        wl.stream.synchronize()
        k_ms = wl.kernel_ms(out, 20)
        verify = wl.verify([out[k].get() for k in ("m0", "m1", "m2")], blk, m, blk.shape[1])
        detail["configs1"] = {"workload": "configs[1]: %dx%dx%d fp32 + uint8 mask, fused moment0+1+2" % c1shape,
                              "value": nz * ny * nx / (k_ms * 1e-3) / 1e6, "unit": "Mvoxel/s", "mask_valid_fraction": valid_frac,
                              "roofline": roofline(wl, k_ms, *pmc_traffic("c2")), "verify": verify}
        if not args.no_configs:
            try:
                detail["next_rows"] = next_rows_records(cube, maskd, blk, m, device)
            except AssertionError as exc:                   # a failed oracle check is reported, never hidden
                detail["next_rows"] = {"error": "oracle check failed: %r" % (exc,)}
        del wl, out, cube, maskd, blk, m
        gc.collect()
        pool_trim(device)
    if not args.no_configs:
        detail["configs"] = config_records(args, device)
        if not args.configs_only:
            try:
                detail["wide"] = wide_records(device, max(1, args.configs_scale))
            except AssertionError as exc:
                detail["wide"] = {"error": "oracle check failed: %r" % (exc,)}
    return detail


def headline_record(shape, device, args):
    """the fixed cube on one GPU: K launches between barriers (wall clock) = `value`, the same measurement the N > 1 lines
    make; the kernel alone by HIP events on its stream = `roofline`; the first rows against the oracle"""
    import gc
    from spectral_cube_amd import synth
    from spectral_cube_amd.device import pool_trim, synchronize
    import numpy as np
    cube, maskd, tile, tmask = tiled_strip_on_device(shape, synth.SEEDS["C4"], device)
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
    k_ms = wl.kernel_ms(out, min(20, max(10, args.steps)))
    verify = wl.verify([out[k].get() for k in ("m0", "m1", "m2")], tile, tmask, 4)
    nz, ny, nx = shape
    rec = {"value": nz * ny * nx * args.steps / elapsed / 1e6, "unit": "Mvoxel/s", "ms_per_step": elapsed / args.steps * 1e3,
           "kernel_only_mvoxel_per_s": nz * ny * nx / (k_ms * 1e-3) / 1e6,
           "mask_valid_fraction": float(np.count_nonzero(tmask)) / tmask.size,
           "roofline": roofline(wl, k_ms, *pmc_traffic("ns")), "verify": verify,
           "verify_compact": {"vs": "oracle (numpy float64 restatement of the reference)", "rows_checked": verify["rows_checked"],
                              "max_scaled_err": sig(max(verify["max_scaled_err_m0_m1_m2"]), 3), "tolerance": 1e-5,
                              "nan_pattern": verify["nan_pattern"]},
           "target_frac": 0.60}
    del wl, out, cube, maskd
    gc.collect()
    pool_trim(device)
    if not getattr(args, "no_strip_terms", False):
        rec["strip_terms"] = strip_terms(shape, device, float(k_ms))
    return rec


def strip_terms(shape, device, t1_ms):
    """the kernel term of the strong-scaling model, MEASURED on one GPU (round-4 verdict, item 6): the headline kernel on the
    row strips a rank would own at N = 2, 4, 8 - a contiguous (nz, ny / N, nx) cube + mask made the way run_sharded makes a
    rank's strip - as one launch and in the four row blocks of distributed.ChunkedMoments (the form that hides the stitch
    inside the call; four launches on blocks of ny / N / 4 rows).  The model of DESIGN 7 assumed T1 / N for it."""
    import gc
    import numpy as np
    from spectral_cube_amd import _lib, ops, synth
    from spectral_cube_amd.device import DeviceArray, Event, pool_trim
    nz, ny, nx = shape
    out = []
    for n in (2, 4, 8):
        rows = ny // n
        if rows < 64 or ny % n:
            continue
        cube, maskd, tile, tmask = tiled_strip_on_device((nz, rows, nx), synth.SEEDS["C4"] + 17, device)
        wl = Workload(cube, maskd, device)
        o = wl.outputs()
        for _ in range(2):
            wl.launch(o)
        wl.stream.synchronize()
        k = wl.kernel_ms(o, 10)
        # the block form: launches on row blocks of the same resident strip, maps per block (as many blocks as
        # distributed.ChunkedMoments picks for this strip height: 4, or 2 at N = 8)
        from spectral_cube_amd.distributed import ChunkedMoments
        nb = ChunkedMoments.pick_chunks(rows)
        rc = rows // nb
        blocks = [(cube.rows(c * rc, (c + 1) * rc), ops.MaskSpec(_lib.MASK_ARRAY, array=maskd.rows(c * rc, (c + 1) * rc)),
                   {q: DeviceArray((rc, nx), np.float64, device) for q in ("m0", "m1", "m2")}) for c in range(nb)]

        def four():
            for cb, mb, ob in blocks:
                ops.moments(cb, wl.d_cen, dv=500.0, m1_add=wl.cref + wl.v[0], mask=mb, want=("m0", "m1", "m2"), stream=wl.stream,
                            workspace=wl.ws, out=ob)
        four(); four()
        wl.stream.synchronize()
        e0, e1, ts = Event(device), Event(device), []
        for _ in range(10):
            e0.record(wl.stream); four(); e1.record(wl.stream); e1.synchronize()
            ts.append(e0.elapsed_ms(e1))
        k4 = Ms(ts)
        alg = nz * rows * nx * 5 + rows * nx * 24
        out.append({"n_gpus_modelled": n, "strip": [nz, rows, nx], "kernel_ms": float(k), "kernel_ms_stats": ms_stats(k),
                    "t1_over_n_ms": t1_ms / n, "kernel_over_t1_over_n": float(k) / (t1_ms / n),
                    "frac": alg / (float(k) * 1e-3) / 1e9 / PEAK_GBS,
                    "blocks": nb, "block_rows": rc, "blocks_ms": float(k4), "blocks_ms_stats": ms_stats(k4),
                    "blocks_over_one_launch": float(k4) / float(k)})
        del wl, o, cube, maskd, blocks
        gc.collect()
        pool_trim(device)
    return out


# ---- BASELINE.json configs[2], [3], [4] at full size (N = 1) ------------------------------------------
def replicate_rows(dev, tile):
    """(nz, ny, nx) device array <- host tile (nz, ty, nx) repeated along y (one upload, D2D doubling)"""
    from spectral_cube_amd import _lib
    nz, ny, nx = dev.shape
    ty, isz = tile.shape[1], dev.dtype.itemsize
    row = nx * isz
    _lib.call("spc_memcpy3d_h2d", dev.device, C.c_void_p(dev.ptr), row, ny * row, tile.ctypes.data_as(C.c_void_p),
              row, ty * row, row, min(ty, ny), nz, None)
    have = min(ty, ny)
    while have < ny:
        n = min(have, ny - have)
        _lib.call("spc_memcpy3d_d2d", dev.device, C.c_void_p(dev.ptr + have * row), row, ny * row,
                  C.c_void_p(dev.ptr), row, ny * row, row, n, nz, None)
        have += n
    _lib.call("spc_device_sync", dev.device)


def replicate_planes(dev, tile):
    """(nz, ny, nx) device array <- host tile (tz, ny, nx) repeated along z"""
    from spectral_cube_amd import _lib
    nz, ny, nx = dev.shape
    tz = min(tile.shape[0], nz)
    plane = ny * nx * dev.dtype.itemsize
    _lib.call("spc_memcpy_h2d", dev.device, C.c_void_p(dev.ptr), tile.ctypes.data_as(C.c_void_p), tz * plane, None)
    have = tz
    while have < nz:
        n = min(have, nz - have)
        _lib.call("spc_memcpy_d2d", dev.device, C.c_void_p(dev.ptr + have * plane), C.c_void_p(dev.ptr), n * plane, None)
        have += n
    _lib.call("spc_device_sync", dev.device)


def fetch_rows(dev, y0, y1):
    """host copy of rows [y0, y1) of every plane of a (nz, ny, nx) device cube"""
    from spectral_cube_amd import _lib
    from spectral_cube_amd.device import DeviceArray
    nz, ny, nx = dev.shape
    row = nx * dev.dtype.itemsize
    tmp = DeviceArray((nz, y1 - y0, nx), dev.dtype, dev.device)
    _lib.call("spc_memcpy3d_d2d", dev.device, C.c_void_p(tmp.ptr), row, (y1 - y0) * row,
              C.c_void_p(dev.ptr + y0 * row), row, ny * row, row, y1 - y0, nz, None)
    return tmp.get()


def event_ms(fn, device, n=10, warm=2):
    """median duration of fn() (Ms: + min / max / mean) by HIP events on the stream its kernels are launched on (the null stream)"""
    import numpy as np
    from spectral_cube_amd.device import Event, synchronize
    for _ in range(warm):
        fn()
    synchronize(device)
    e0, e1 = Event(device), Event(device)
    ts = []
    for _ in range(n):
        e0.record(None)
        fn()
        e1.record(None)
        e1.synchronize()
        ts.append(e0.elapsed_ms(e1))
    return Ms(ts)


def cfg_record(key, arithmetic, name, kernel, ms, alg_bytes, voxels, verify, bytes_note, **extra):
    """one timed record: `key` = its short stable name (compact line, PMC file), `arithmetic` = what the timed kernel computes in"""
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    traffic, src = pmc_traffic(key)
    rec = {"key": key, "name": name, "kernel": kernel, "arithmetic": arithmetic, "kernel_ms": float(ms), "kernel_ms_stats": ms_stats(ms),
           "algorithmic_bytes": int(alg_bytes), "bytes_per_voxel": bytes_note,
           "achieved_GBps": gbs, "frac": gbs / PEAK_GBS, "traffic": traffic, "traffic_source": src,
           "traffic_over_algorithmic": (traffic / alg_bytes) if traffic else None,
           "value": voxels / (ms * 1e-3) / 1e6, "unit": "Mvoxel/s", "verify": verify}
    rec.update(extra)
    return rec


def _close(got, exp, scale, what, tol=1e-5):
    """NaN pattern identical, |got - exp| <= tol * scale; returns the scaled error"""
    import numpy as np
    got, exp = np.asarray(got, dtype=np.float64), np.asarray(exp, dtype=np.float64)
    assert np.array_equal(np.isnan(got), np.isnan(exp)), what + ": NaN pattern mismatch vs oracle"
    ok = ~np.isnan(exp)
    err = float(np.abs(got[ok] - exp[ok]).max() / scale) if ok.any() else 0.0
    assert err <= tol, (what, err)
    return err


def config_c3(device, scale):
    """configs[2]: 2048^3 fp32, spectral_smooth(Gaussian sigma = 4 channels: 33 taps) then moment1.
    Rows repeat a seeded 2-row tile.  All valid (the config as written) and with a uint8 mask."""
    import numpy as np
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import oracle_np as O
    from spectral_cube_amd import _lib, ops, synth, Gaussian1DKernel
    from spectral_cube_amd.device import DeviceArray
    nz, ny, nx = 2048, 2048 // scale, 2048
    vox = nz * ny * nx
    tile = synth.gaussian_line_cube((nz, 2, nx), synth.SEEDS["C3"], chunk_rows=2)
    tmask = synth.boolean_mask(tile, synth.SEEDS["C3"])
    cube = DeviceArray((nz, ny, nx), np.float32, device)
    replicate_rows(cube, tile)
    k = Gaussian1DKernel(4).array
    v = synth.spectral_axis(nz)
    cen = v - v[0]
    cref = cen[nz // 2]
    d_cen = DeviceArray.from_numpy(cen - cref, device)
    o1 = {"m1": DeviceArray((ny, nx), np.float64, device)}
    W = 256                                                   # columns the oracle redoes
    recs = []

    def check_m1(inc, what):
        got = o1["m1"].get()
        sm = O.spectral_smooth(tile[:, :, :W], inc, k)
        exp = O.moment(sm, inc, 1, cen, 500.0, world0=v[0])
        s0 = O.moment(sm, inc, 0, cen, 1.0)
        with np.errstate(invalid="ignore"):
            wc = np.abs(s0) > 5.0                             # moment 1 = S1 / S0: well-conditioned spaxels
        assert np.array_equal(np.isnan(got[:2, :W]), np.isnan(exp)), what + ": NaN pattern"
        err = float(np.abs(got[:2, :W][wc] - exp[wc]).max() / (500.0 * nz))
        assert err <= 1e-5, (what, err)
        assert np.array_equal(got[-2:], got[:2], equal_nan=True), what + ": not periodic in y"
        return {"max_scaled_err": err, "spaxels_checked": int(wc.sum()), "rows_periodic": True}

    def check_cube(out, inc, what):
        got = fetch_rows(out, ny - 2, ny)[:, :, :W]
        exp = O.spectral_smooth(tile[:, :, :W], inc, k)
        return {"max_scaled_err": _close(got, exp, float(np.nanmax(np.abs(exp))), what), "voxels_checked": int(exp.size)}

    ms = event_ms(lambda: ops.spectral_conv_moments(cube, k, d_cen, dv=500.0, m1_add=cref + v[0], want=("m1",), out=o1,
                                                    cen_host=cen - cref), device)
    recs.append(cfg_record("c3_fused", A_SPEC_ALG, "C3 spectral_smooth(33 taps) -> moment1, fused, all valid", "weighted_moments_kernel (algebraic fusion)",
                           ms, vox * 4 + ny * nx * 8, vox, check_m1(None, "C3 fused"), "4 read + 8 B/spaxel out"))
    sm = DeviceArray((nz, ny, nx), np.float32, device)
    ms = event_ms(lambda: ops.spectral_conv(cube, k, out=sm), device)
    recs.append(cfg_record("c3_mat", A_SPEC, "C3 spectral_smooth(33 taps) materialised, all valid", "spectral_conv_fast_kernel<33>", ms, vox * 8, vox,
                           check_cube(sm, None, "C3 smooth"), "4 read + 4 written"))
    maskd = DeviceArray((nz, ny, nx), np.uint8, device)
    replicate_rows(maskd, tmask)
    mspec = ops.MaskSpec(_lib.MASK_ARRAY, array=maskd)
    inc = tmask[:, :, :W].astype(bool)
    ms = event_ms(lambda: ops.spectral_conv_moments(cube, k, d_cen, dv=500.0, m1_add=cref + v[0], mask=mspec, want=("m1",),
                                                    out=o1, cen_host=cen - cref), device)
    recs.append(cfg_record("c3_fused_mask", A_SPEC + "; f64 moment sums", "C3 spectral_smooth(33 taps) -> moment1, fused, uint8 mask", "spectral_conv_kernel<33,true,true,false,true> (ARR, FUSE, SYM)", ms,
                           vox * 5 + ny * nx * 8, vox, check_m1(inc, "C3 fused masked"), "4 + 1 read + 8 B/spaxel out",
                           mask_valid_fraction=float(tmask.mean())))
    ms = event_ms(lambda: ops.spectral_conv(cube, k, mask=mspec, out=sm), device)
    recs.append(cfg_record("c3_mat_mask", A_SPEC, "C3 spectral_smooth(33 taps) materialised, uint8 mask", "spectral_conv_kernel<33,true,false,false,true> (ARR, SYM)", ms, vox * 9, vox,
                           check_cube(sm, inc, "C3 smooth masked"), "4 + 1 read + 4 written",
                           mask_valid_fraction=float(tmask.mean())))
    return recs


def config_c4(device, scale):
    """configs[3]: 4096 x 2048 x 2048 fp32 (+ uint8 mask), spatial_smooth with a 2-D Gaussian of FWHM = 8 px
    (29 x 29 taps) + moment0.  Planes repeat a seeded 2-plane tile."""
    import numpy as np
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import oracle_np as O
    from spectral_cube_amd import _lib, ops, synth, Gaussian2DKernel
    from spectral_cube_amd.device import DeviceArray
    nz, ny, nx = 4096 // scale, 2048, 2048
    vox = nz * ny * nx
    rng = np.random.default_rng(synth.SEEDS["C4"])
    tile = rng.standard_normal((2, ny, nx), dtype=np.float32) + 2.0
    tmask = (rng.random((2, ny, nx), dtype=np.float32) > 0.2).view(np.uint8)
    tmask[:, :8, :8] = 0
    cube = DeviceArray((nz, ny, nx), np.float32, device)
    replicate_planes(cube, tile)
    k2 = Gaussian2DKernel(8 / 2.3548200450309493).array
    assert k2.shape == (29, 29)
    sm = DeviceArray((nz, ny, nx), np.float32, device)
    WY, WX, PAD = 96, 160, 14
    sub = (slice(None), slice(0, WY + PAD), slice(0, WX + PAD))
    win = (slice(None), slice(0, WY), slice(0, WX))
    recs = []
    # oracle windows (round 5): the origin corner, one in the middle of the map (no plane edge: rows 1000.., columns 900.. -
    # not multiples of the kernels' tiles) and the FAR corner (last rows and columns: the partial-tile and clamping code)
    WINDOWS = ((0, 0), (1000, 900), (ny - WY, nx - WX))

    def window_slices(y0, x0):
        ya, yb, xa, xb = max(y0 - PAD, 0), min(y0 + WY + PAD, ny), max(x0 - PAD, 0), min(x0 + WX + PAD, nx)
        return (slice(None), slice(ya, yb), slice(xa, xb)), (slice(None), slice(y0 - ya, y0 - ya + WY), slice(x0 - xa, x0 - xa + WX))

    def smoothed_window(out, y0=0, x0=0):
        got = np.empty((2, WY, WX), np.float32)
        for z in range(2):                                   # the LAST two planes of the cube
            rows = np.empty((WY, nx), np.float32)
            _lib.call("spc_memcpy_d2h", device, rows.ctypes.data_as(C.c_void_p),
                      C.c_void_p(out.ptr + ((nz - 2 + z) * ny + y0) * nx * 4), rows.nbytes, None)
            got[z] = rows[:, x0:x0 + WX]
        return got

    def check_cube_windows(out, inc_tile, what):
        """smoothed cube against the oracle in the three windows; returns the verify record"""
        worst = 0.0
        for y0, x0 in WINDOWS:
            s_, w_ = window_slices(y0, x0)
            e_ = O.spatial_smooth(tile[s_], None if inc_tile is None else inc_tile[s_].astype(bool), k2)[w_]
            worst = max(worst, _close(smoothed_window(out, y0, x0), e_, float(np.nanmax(np.abs(e_))), "%s window (%d, %d)" % (what, y0, x0)))
        return {"max_scaled_err": worst, "voxels_checked": int(2 * WY * WX * len(WINDOWS)), "windows": [list(w) for w in WINDOWS]}

    def check_m0_windows(m0map, inc_tile, dv, what):
        """moment 0 of the smoothed cube under the original mask against the oracle in the three windows"""
        worst = 0.0
        for y0, x0 in WINDOWS:
            s_, w_ = window_slices(y0, x0)
            e_ = O.spatial_smooth(tile[s_], None if inc_tile is None else inc_tile[s_].astype(bool), k2)[w_]
            if inc_tile is None:
                em = (nz // 2) * dv * e_.astype(np.float64).sum(axis=0)
            else:
                iw = inc_tile[:, y0:y0 + WY, x0:x0 + WX].astype(bool)
                em = (nz // 2) * dv * np.where(iw, e_, 0.0).sum(axis=0)
                em[~iw.any(axis=0)] = np.nan
            worst = max(worst, _close(np.asarray(m0map)[y0:y0 + WY, x0:x0 + WX], em, float(np.nanmax(np.abs(em))), "%s window (%d, %d)" % (what, y0, x0)))
        return {"max_scaled_err": worst, "spaxels_checked": int(WY * WX * len(WINDOWS)), "windows": [list(w) for w in WINDOWS]}

    ms = event_ms(lambda: ops.spatial_conv(cube, k2, out=sm), device, n=5, warm=1)
    ver = check_cube_windows(sm, None, "C4 smooth")
    recs.append(cfg_record("c4_mat", A_SPAT_F32, "C4 spatial_smooth(29x29), all valid", "spatial_sep_fast_kernel<29>", ms, vox * 8, vox, ver, "4 read + 4 written"))

    maskd = DeviceArray((nz, ny, nx), np.uint8, device)
    replicate_planes(maskd, tmask)
    mspec = ops.MaskSpec(_lib.MASK_ARRAY, array=maskd)
    inc = tmask[sub].astype(bool)
    os.environ["SPC_SPATIAL_RING"] = "1"                     # (the entry point takes the split form first since round 5: this record is the ring kernel)
    ms_s = event_ms(lambda: ops.spatial_conv(cube, k2, mask=mspec, out=sm), device, n=5, warm=1)
    ver = check_cube_windows(sm, tmask, "C4 smooth masked")
    recs.append(cfg_record("c4_mat_mask_ring", A_SPAT_F32, "C4 spatial_smooth(29x29), uint8 mask (ring kernel, vector ALU)", "spatial_sep_grouped_kernel<29,true,false,true,256,true,0> (ARR, ISO, 256 threads, SYM)", ms_s, vox * 9, vox, ver,
                           "4 + 1 read + 4 written", mask_valid_fraction=float(tmask.mean())))
    # the same cube -> cube operator in the split form (every product on the fp16 matrix instruction)
    ms_x = event_ms(lambda: ops.spatial_conv_mfma(cube, k2, mask=mspec, out=sm), device, n=5, warm=1)
    ver = check_cube_windows(sm, tmask, "C4 smooth masked, split form")
    recs.append(cfg_record("c4_mat_mask", A_SPAT_SPLIT, "C4 spatial_smooth(29x29), uint8 mask, matrix cores (fp16 hi / lo split form)", "spatial_split_kernel<3,0,true,2,true,0> (ARR, STORE; whole-column march)", ms_x, vox * 9, vox, ver,
                           "4 + 1 read + 4 written", mask_valid_fraction=float(tmask.mean())))
    del os.environ["SPC_SPATIAL_RING"]

    # the pipeline of the config: spatial_smooth -> moment0 (the smoothed cube keeps the ORIGINAL mask)
    cen = DeviceArray.from_numpy(np.zeros(nz), device)
    o0 = {"m0": DeviceArray((ny, nx), np.float64, device)}
    need = _lib.load().spc_moments_workspace_bytes(nz, ny, nx)
    ws = DeviceArray((max(need, 1),), np.uint8, device)

    def pipeline_masked():
        ops.spatial_conv(cube, k2, mask=mspec, out=sm)
        ops.moments(sm, cen, dv=500.0, mask=mspec, want=("m0",), out=o0, workspace=ws)
    ms = event_ms(pipeline_masked, device, n=5, warm=1)
    ver = check_m0_windows(o0["m0"].get(), tmask, 500.0, "C4 moment0 masked")
    recs.append(cfg_record("c4_pipe_mask", A_SPAT_SPLIT + "; f64 moment sums", "C4 pipeline spatial_smooth(29x29) -> moment0, uint8 mask (materialised)",
                           "spatial_split_kernel<3,0,true,2,true,0> + moments_kernel", ms, vox * 5 + ny * nx * 8, vox, ver,
                           "fused ideal: 4 + 1 read + 8 B/spaxel out (the materialised form moves 9 + 5 B/voxel)",
                           mask_valid_fraction=float(tmask.mean())))

    # the same pipeline FUSED: numerator and denominator of the NaN-aware convolution on the matrix cores, moment sums kept on
    # chip, the smoothed cube never written (spc_spatial_conv_sep_mfma_f32)
    m0f = DeviceArray((ny, nx), np.float64, device)
    ms = event_ms(lambda: ops.spatial_conv_mfma(cube, k2, mask=mspec, want_cube=False, want_m0=True, dv=500.0, m0=m0f), device, n=5, warm=1)
    ver = check_m0_windows(m0f.get(), tmask, 500.0, "C4 fused moment0 masked")
    recs.append(cfg_record("c4_fused_mask", A_SPAT_SPLIT + "; f32 channel-chunk sums, f64 map", "C4 pipeline spatial_smooth(29x29) -> moment0, uint8 mask, FUSED (matrix cores, cube never written)",
                           "spatial_split_kernel<3,4,true,2,false,1> (ARR, sums) (+ split_finish_kernel)", ms, vox * 5 + ny * nx * 8, vox, ver,
                           "4 + 1 read + 8 B/spaxel out", mask_valid_fraction=float(tmask.mean())))
    # moments 1 and 2 of the smoothed cube from the same kernel (three sums per spaxel instead of one)
    cen_np = (np.arange(nz) - nz // 2) * 500.0
    d_cen = DeviceArray.from_numpy(cen_np, device)
    mm = {}

    def fused012():
        mm.update(ops.spatial_conv_mfma_moments(cube, k2, d_cen, dv=500.0, m1_add=0.0, mask=mspec)[1])
    ms = event_ms(fused012, device, n=5, warm=1)
    worst = check_m0_windows(mm["m0"].get(), tmask, 500.0, "C4 fused moments: moment0")["max_scaled_err"]
    g1, g2 = mm["m1"].get(), mm["m2"].get()
    for y0, x0 in WINDOWS:                                   # planes alternate between the two tile planes: sums in closed form
        s_, w_ = window_slices(y0, x0)
        e_ = O.spatial_smooth(tile[s_], tmask[s_].astype(bool), k2)[w_].astype(np.float64)
        iw = tmask[:, y0:y0 + WY, x0:x0 + WX].astype(bool)
        f_ = np.where(iw, e_, 0.0)
        c0, c1 = cen_np[0::2], cen_np[1::2]
        s0 = (nz // 2) * (f_[0] + f_[1])
        s1 = f_[0] * c0.sum() + f_[1] * c1.sum()
        s2 = f_[0] * (c0 ** 2).sum() + f_[1] * (c1 ** 2).sum()
        with np.errstate(all="ignore"):
            e1, e2 = s1 / s0, s2 / s0 - (s1 / s0) ** 2
        worst = max(worst, _close(g1[y0:y0 + WY, x0:x0 + WX], e1, 500.0 * nz, "C4 fused moment1 window (%d, %d)" % (y0, x0)))
        worst = max(worst, _close(g2[y0:y0 + WY, x0:x0 + WX], e2, float(np.nanmax(np.abs(e2))), "C4 fused moment2 window (%d, %d)" % (y0, x0)))
    recs.append(cfg_record("c4_fused012_mask", A_SPAT_SPLIT + "; " + A_FUSED012, "C4 pipeline spatial_smooth(29x29) -> moment0 + moment1 + moment2, uint8 mask, FUSED (matrix cores)",
                           "spatial_split_kernel<3,6,true,2,false,3> (ARR, three sums, eight channel-parallel waves) (+ split_finish_kernel)", ms, vox * 5 + ny * nx * 24, vox,
                           {"max_scaled_err": worst, "spaxels_checked": int(3 * WY * WX * len(WINDOWS)), "windows": [list(w) for w in WINDOWS]},
                           "4 + 1 read + 24 B/spaxel out", mask_valid_fraction=float(tmask.mean())))
    del mm, g1, g2
    # a SIGNAL mask instead of the 80 % random one: coherent regions (35 % valid), what `data > 2 sigma` of a real cube looks like -
    # the stencils see long runs of all-valid / all-invalid windows
    yy_, xx_ = np.mgrid[0:ny, 0:nx]
    smask = np.stack([(np.sin(xx_ / 37.0) * np.cos(yy_ / 53.0) > 0.2), (np.sin(xx_ / 41.0 + 1.0) * np.cos(yy_ / 47.0) > 0.2)]).view(np.uint8)
    replicate_planes(maskd, smask)
    os.environ["SPC_SPATIAL_RING"] = "1"
    ms_sig = event_ms(lambda: ops.spatial_conv(cube, k2, mask=mspec, out=sm), device, n=5, warm=1)
    del os.environ["SPC_SPATIAL_RING"]
    ver = check_cube_windows(sm, smask, "C4 smooth signal mask")
    recs.append(cfg_record("c4_mat_sigmask_ring", A_SPAT_F32, "C4 spatial_smooth(29x29), uint8 SIGNAL mask (coherent regions)", "spatial_sep_grouped_kernel<29,true,false,true,256,true,0>",
                           ms_sig, vox * 9, vox, ver, "4 + 1 read + 4 written", mask_valid_fraction=float(smask.mean())))
    ms = event_ms(lambda: ops.spatial_conv_mfma(cube, k2, mask=mspec, want_cube=False, want_m0=True, dv=500.0, m0=m0f), device, n=5, warm=1)
    ver = check_m0_windows(m0f.get(), smask, 500.0, "C4 fused moment0 signal mask")
    recs.append(cfg_record("c4_fused_sigmask", A_SPAT_SPLIT + "; f32 channel-chunk sums, f64 map", "C4 pipeline spatial_smooth(29x29) -> moment0, uint8 SIGNAL mask, FUSED (matrix cores)",
                           "spatial_split_kernel<3,4,true,2,false,1> (ARR, sums) (+ split_finish_kernel)", ms, vox * 5 + ny * nx * 8, vox, ver,
                           "4 + 1 read + 8 B/spaxel out", mask_valid_fraction=float(smask.mean())))

    # all valid: convolution commutes with the sums along z (algebraic path of SpectralCube.spatial_smooth -> moment0)
    from spectral_cube_amd import SpectralCube
    del sm, maskd, mspec
    hdr = {"NAXIS": 3, "NAXIS1": nx, "NAXIS2": ny, "NAXIS3": nz, "CTYPE1": "RA---TAN", "CTYPE2": "DEC--TAN", "CTYPE3": "VRAD",
           "CRVAL3": 0.0, "CDELT3": 500.0, "CRPIX3": 1.0, "CUNIT3": "m/s", "CDELT1": -1e-4, "CDELT2": 1e-4, "CRPIX1": 1.0,
           "CRPIX2": 1.0, "CRVAL1": 10.0, "CRVAL2": 20.0, "BUNIT": "K"}
    sc = SpectralCube.from_device(cube, header=hdr)
    kobj = Gaussian2DKernel(8 / 2.3548200450309493)
    res = {}

    def cube_level():
        res["m0"] = sc.spatial_smooth(kobj).moment0()
    # HIP events on the null stream around the whole cube-level call (kernels, the device-side map checks, 32 MiB to the host):
    # median of 7 - one wall-clock sample stood here in round 4
    ms = event_ms(cube_level, device, n=7, warm=2)
    ver = check_m0_windows(np.asarray(res["m0"]), None, 500.0, "C4 moment0 all valid")
    recs.append(cfg_record("c4_fused", "f64 sums along z, then the 29x29 map convolution in f64", "C4 pipeline spatial_smooth(29x29) -> moment0, all valid (the cube-level call, events around it)",
                           "moments_kernel + map_conv2d (algebraic: conv commutes with the z sums)", ms, vox * 4 + ny * nx * 8, vox,
                           ver, "4 read + 8 B/spaxel out", timing="HIP events on the null stream around the cube-level call (host work between its kernels included), median of 7"))
    return recs


def config_c5(device, scale):
    """configs[4]: 2048 x 1024 x 1024 cube, spectral_interpolate to 4096 channels, then reproject onto the same
    TAN grid rotated by 30 degrees."""
    import numpy as np
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import oracle_np as O
    from spectral_cube_amd import ops, synth
    from spectral_cube_amd.device import DeviceArray
    from spectral_cube_amd.wcs import SimpleWCS
    nz, ny, nx = 2048, 1024 // scale, 1024
    nzo = 4096
    tile = synth.gaussian_line_cube((nz, 2, nx), synth.SEEDS["C5"], chunk_rows=2)
    tile[700, 1, 9] = np.nan
    cube = DeviceArray((nz, ny, nx), np.float32, device)
    replicate_rows(cube, tile)
    v = synth.spectral_axis(nz)
    grid = np.linspace(v[0], v[-1], nzo)
    lo, t, inv, _, _, fill = ops.lerp_plan(v, grid)
    out = DeviceArray((nzo, ny, nx), np.float32, device)
    recs = []
    ms = event_ms(lambda: ops.spectral_lerp(cube, lo, t, inv, fill, out=out), device)
    exp, _ = O.spectral_interpolate(tile, None, v, grid)
    got = fetch_rows(out, ny - 2, ny)
    ver = {"max_scaled_err": _close(got, exp, float(np.nanmax(np.abs(exp))), "C5 lerp"), "voxels_checked": int(exp.size)}
    recs.append(cfg_record("c5_lerp", A_LERP, "C5 spectral_interpolate 2048 -> 4096 channels", "spectral_lerp_tiles_kernel<4>", ms, (nz + nzo) * ny * nx * 4,
                           nzo * ny * nx, ver, "4 B x nz_in + 4 B x nz_out per spaxel"))
    # reproject: rotated TAN header, device pixel map, bilinear
    hdr = {"CTYPE1": "RA---TAN", "CTYPE2": "DEC--TAN", "CRVAL1": 150.0, "CRVAL2": 2.0, "CRPIX1": nx / 2 + 0.5, "CRPIX2": ny / 2 + 0.5,
           "CDELT1": -1 / 3600, "CDELT2": 1 / 3600, "NAXIS": 2, "NAXIS1": nx, "NAXIS2": ny}
    c, s_ = np.cos(np.radians(30)), np.sin(np.radians(30))
    w_in, w_out = SimpleWCS(hdr, naxis=2), SimpleWCS(dict(hdr, PC1_1=c, PC1_2=-s_, PC2_1=s_, PC2_2=c), naxis=2)
    xs, ys = ops.wcs_pixel_map(w_in, w_out, (ny, nx), device)
    rep = DeviceArray((nzo, ny, nx), np.float32, device)
    ms = event_ms(lambda: ops.resample_bilinear(out, xs, ys, out=rep, want_footprint=False), device, n=5, warm=1)
    hx, hy = xs.get(), ys.get()
    chans = [0, 1, nzo // 2 + 1, nzo - 1]
    src = np.stack([out.planes(c_, c_ + 1).get()[0] for c_ in chans])
    exp, _ = O.resample_bilinear(src, hx, hy)
    got = np.stack([rep.planes(c_, c_ + 1).get()[0] for c_ in chans])
    ver = {"max_scaled_err": _close(got, exp, float(np.nanmax(np.abs(src))), "C5 reproject"), "voxels_checked": int(exp.size),
           "pixel_map": "spc_wcs_pixel_map_f64 on the device"}
    recs.append(cfg_record("c5_reproject", A_BIL, "C5 reproject 4096 x 1024^2 onto the grid rotated by 30 deg (bilinear)", "bilinear_lds_kernel<64>", ms,
                           nzo * ny * nx * 8, nzo * ny * nx, ver, "~4 read + 4 written per output voxel"))
    # the pipeline in ONE pass (what cube.spectral_interpolate(grid).reproject(header) runs): every input plane resampled once,
    # output channels blended from neighbouring resampled planes in the same kernel; checked against the oracle's resampling of
    # the (oracle-checked) interpolated planes above
    ms = event_ms(lambda: ops.resample_bilinear_lerp(cube, xs, ys, lo, t, inv, out=rep, want_footprint=False), device, n=5, warm=1)
    got = np.stack([rep.planes(c_, c_ + 1).get()[0] for c_ in chans])
    ver = {"max_scaled_err": _close(got, exp, float(np.nanmax(np.abs(src))), "C5 one pass"), "voxels_checked": int(exp.size),
           "pixel_map": "spc_wcs_pixel_map_f64 on the device"}
    recs.append(cfg_record("c5_one_pass", A_BIL + "; " + A_LERP, "C5 pipeline spectral_interpolate 2048 -> 4096 channels -> reproject (rotated 30 deg), ONE pass (the interpolated cube is never formed)",
                           "bilinear_lds_kernel<64, LERP>", ms, (nz + nzo) * ny * nx * 4, nzo * ny * nx, ver,
                           "4 B x nz_in read + 4 B x nz_out written per spaxel"))
    del cube
    return recs


def next_rows_records(cube, maskd, tile, tmask, device):
    """SURVEY.md section 8(f) rows timed on the configs[1] cube while it is resident (kernel time by HIP events, algorithmic
    bytes, fraction of 8 TB/s), each checked against the oracle / numpy on the cube's first rows: f1 statistics(), f4 the median
    along the spectral axis (bit-exact) and sigma_clip_spectrally (astropy defaults, 3 sigma)."""
    import warnings
    import numpy as np
    from spectral_cube_amd import _lib, ops
    from spectral_cube_amd.device import DeviceArray
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import oracle_np as O
    nz, ny, nx = cube.shape
    vox = nz * ny * nx
    rows = tile.shape[1]
    spec = ops.MaskSpec(_lib.MASK_ARRAY, array=maskd)
    inc = tmask.astype(bool) & ~np.isnan(tile)
    out = {}
    # f1: one pass, five statistics
    ms = event_ms(lambda: ops.stats_global(cube, mask=spec), device)
    st = ops.stats_global(cube.rows(0, rows), mask=spec.rows(0, rows))
    sel = tile[inc].astype(np.float64)
    assert st["npts"] == sel.size and st["min"] == sel.min() and st["max"] == sel.max(), "statistics(): count / extrema"
    assert abs(st["sum"] - sel.sum()) <= 1e-10 * np.abs(sel).sum() and abs(st["sumsq"] - (sel * sel).sum()) <= 1e-10 * (sel * sel).sum()
    out["f1_statistics"] = cfg_record("f1_stats", A_STATS, "f1 statistics(): npts / min / max / sum / sumsq in one pass, 1024^3 + uint8 mask", "stats_global_kernel<ARR> (followed by stats_finish_kernel; the timed call also waits for the 40-byte record on the host)", ms,
                                      vox * 5, vox, {"rows_checked": rows, "npts_min_max": "exact", "sum_sumsq_rel_err": "<= 1e-10"},
                                      "4 B data + 1 B mask read per voxel")
    # f4: median along the spectral axis, rays resident in registers
    med = DeviceArray((ny, nx), np.float32, device)
    ms = event_ms(lambda: ops.percentile_axis0(cube, 50.0, mask=spec, out=med), device)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = np.nanmedian(np.where(inc, tile, np.nan).astype(np.float32), axis=0)
    assert np.array_equal(med.get()[:rows], exp, equal_nan=True), "median(axis=0) differs from np.nanmedian"
    out["f4_median_axis0"] = cfg_record("f4_median", A_SELECT, "f4 median(axis=0), 1024^3 + uint8 mask", "select_reg_kernel<32,64,ARR,DESC,512> (one read of the cube)",
                                        ms, vox * 5 + ny * nx * 4, vox, {"rows_checked": rows, "vs_np_nanmedian": "bit-identical"},
                                        "4 B data + 1 B mask read per voxel, one float32 map out")
    # f4: sigma clipping, the whole loop in one kernel
    keep = {}

    def clip():
        keep["r"] = None                                    # (the previous result goes back to the pool first)
        keep["r"] = ops.sigma_clip_axis0(cube, sigma=3.0, mask=spec)
    ms = event_ms(clip, device, n=5, warm=1)
    got = fetch_rows(keep["r"], 0, rows)
    exp = O.sigma_clip(tile, inc, 3.0)
    differ = float(np.mean(np.isnan(got) != np.isnan(exp)))
    both = ~np.isnan(got) & ~np.isnan(exp)
    assert differ < 2e-4 and np.array_equal(got[both], exp[both]), ("sigma clip vs oracle", differ)
    keep.clear()
    valid = float(np.mean(inc))
    out["f4_sigma_clip"] = cfg_record("f4_clip", A_CLIP, "f4 sigma_clip_spectrally(3), astropy defaults (median / std, <= 5 iterations), 1024^3 + uint8 mask (the cube's own signal mask: %.1f %% valid)" % (100 * valid),
                                      "sigma_clip_reg_kernel<32,64,true,false,true,512> (ARR, std, DESC; rays of <= 128 valid samples packed, one wave each; preceded by clip_probe_kernel, followed by an empty 256-thread grid; the timed call also takes its 4 GiB result from the pool)", ms, vox * 9, vox,
                                      {"rows_checked": rows, "clipped_set_vs_oracle": "identical up to %.1e of the samples (float32 bounds)" % max(differ, 0.0),
                                       "kept_values": "bit-identical"},
                                      "4 B data + 1 B mask read, 4 B written per voxel")
    out["f4_sigma_clip"]["mask_valid_fraction"] = valid
    out["f4_median_axis0"]["mask_valid_fraction"] = valid
    # the same two operators under a DENSE mask (80 % of the samples valid, random): the signal mask above leaves ~50 samples per
    # ray, which the clip kernel packs; here every ray stays in the registers of its block
    rng = np.random.default_rng(4242)
    dense_tile = (rng.random((61, ny, nx), dtype=np.float32) < 0.8).view(np.uint8)
    dense = DeviceArray((nz, ny, nx), np.uint8, device)
    replicate_planes(dense, dense_tile)
    dspec = ops.MaskSpec(_lib.MASK_ARRAY, array=dense)
    ms = event_ms(lambda: ops.percentile_axis0(cube, 50.0, mask=dspec, out=med), device)
    dmask = np.concatenate([dense_tile[:, :rows]] * (nz // 61 + 1))[:nz].astype(bool)
    dinc = dmask & ~np.isnan(tile)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = np.nanmedian(np.where(dinc, tile, np.nan).astype(np.float32), axis=0)
    assert np.array_equal(med.get()[:rows], exp, equal_nan=True), "median(axis=0), dense mask, differs from np.nanmedian"
    out["f4_median_axis0_dense"] = cfg_record("f4_median_dense", A_SELECT, "f4 median(axis=0), 1024^3 + uint8 mask, 80 % valid (random)", "select_reg_kernel<32,64,true,true,512> (one read of the cube)",
                                              ms, vox * 5 + ny * nx * 4, vox, {"rows_checked": rows, "vs_np_nanmedian": "bit-identical"},
                                              "4 B data + 1 B mask read per voxel, one float32 map out")
    out["f4_median_axis0_dense"]["mask_valid_fraction"] = float(np.mean(dinc))

    def clipd():
        keep["r"] = None
        keep["r"] = ops.sigma_clip_axis0(cube, sigma=3.0, mask=dspec)
    ms = event_ms(clipd, device, n=5, warm=1)
    got = fetch_rows(keep["r"], 0, rows)
    exp = O.sigma_clip(tile, dinc, 3.0)
    differ = float(np.mean(np.isnan(got) != np.isnan(exp)))
    both = ~np.isnan(got) & ~np.isnan(exp)
    assert differ < 2e-4 and np.array_equal(got[both], exp[both]), ("sigma clip, dense mask, vs oracle", differ)
    keep.clear()
    out["f4_sigma_clip_dense"] = cfg_record("f4_clip_dense", A_CLIP, "f4 sigma_clip_spectrally(3), astropy defaults, 1024^3 + uint8 mask, 80 % valid (random)",
                                            "sigma_clip_reg_kernel<16,64,true,false,true,256> (ARR, std, DESC; the rays stay in the registers of their block; preceded by clip_probe_kernel and an empty 512-thread grid)", ms, vox * 9, vox,
                                            {"rows_checked": rows, "clipped_set_vs_oracle": "identical up to %.1e of the samples (float32 bounds)" % max(differ, 0.0),
                                             "kept_values": "bit-identical"},
                                            "4 B data + 1 B mask read, 4 B written per voxel")
    out["f4_sigma_clip_dense"]["mask_valid_fraction"] = float(np.mean(dinc))
    del dense, dspec
    return out


def wide_records(device, scale=1):
    """the float64 operators (spc_wide_ops.hip / spc_moments_f64.hip: a float64 source stays float64, masks.py:225) at
    512 x 1024 x 1024 float64 (4.3 GB) + uint8 mask (80 % valid, random): kernel time by HIP events, algorithmic bytes,
    each checked against the float64 oracle / numpy on the first rows (the rows repeat a seeded 8-row tile)."""
    import warnings
    import numpy as np
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import oracle_np as O
    from spectral_cube_amd import _lib, ops, Gaussian1DKernel, Gaussian2DKernel
    from spectral_cube_amd.device import DeviceArray
    nz, ny, nx = 512, 1024 // scale, 1024
    vox = nz * ny * nx
    TR = 8
    rng = np.random.default_rng(4711)
    tile = 1000.0 + rng.standard_normal((nz, TR, nx))
    tmask = (rng.random(tile.shape) < 0.8).view(np.uint8)
    cube, maskd = DeviceArray((nz, ny, nx), np.float64, device), DeviceArray((nz, ny, nx), np.uint8, device)
    replicate_rows(cube, tile)
    replicate_rows(maskd, tmask)
    spec = ops.MaskSpec(_lib.MASK_ARRAY, array=maskd)
    inc = tmask.astype(bool)
    out = DeviceArray((nz, ny, nx), np.float64, device)
    A64 = "f64 samples, f64 arithmetic (the reference keeps a float64 source in float64)"
    recs = []

    def rel(got, exp, what, tol=1e-12):
        got, exp = np.asarray(got, dtype=np.float64), np.asarray(exp, dtype=np.float64)
        assert np.array_equal(np.isnan(got), np.isnan(exp)), what + ": NaN pattern mismatch vs oracle"
        ok = ~np.isnan(exp)
        err = float(np.abs(got[ok] - exp[ok]).max() / np.abs(exp[ok]).max()) if ok.any() else 0.0
        assert err <= tol, (what, err)
        return err

    # statistics(): one pass
    ms = event_ms(lambda: ops.stats_global_f64(cube, mask=spec), device)
    st = ops.stats_global_f64(cube.rows(0, TR), mask=spec.rows(0, TR))
    sel = tile[inc]
    assert st["npts"] == sel.size and st["min"] == sel.min() and st["max"] == sel.max() and abs(st["sum"] - sel.sum()) <= 1e-12 * abs(sel.sum())
    recs.append(cfg_record("w_stats_f64", A64, "float64 statistics(): npts / min / max / sum / sumsq in one pass, 512x1024x1024 f64 + uint8 mask",
                           "stats64_global_kernel (+ finish; the timed call waits for the 40-byte record)", ms, vox * 9, vox,
                           {"rows_checked": TR, "npts_min_max": "exact", "sum_rel_err": "<= 1e-12"}, "8 B data + 1 B mask read per voxel"))
    # moment 0 / 1 (+ the second pass of moment 2)
    cen = (np.arange(nz) - nz // 2) * 500.0
    d_cen = DeviceArray.from_numpy(cen, device)
    res = {}

    def mom():
        res.update(ops.moments_f64(cube, d_cen, dv=500.0, m1_add=0.0, mask=spec, want=("m0", "m1", "m2")))
    ms = event_ms(mom, device)
    e = [O.moment(tile, inc, o, cen, 500.0, world0=0.0) for o in (0, 1, 2)]
    err = max(rel(res["m0"].get()[:TR], e[0], "f64 moment0"), rel(res["m1"].get()[:TR], e[1], "f64 moment1", 1e-9),
              rel(res["m2"].get()[:TR], e[2], "f64 moment2", 1e-9))
    recs.append(cfg_record("w_moments_f64", A64 + "; moment 2 as a second pass about moment 1 (the reference's own form)",
                           "float64 moment0 + moment1 + moment2, 512x1024x1024 f64 + uint8 mask (two passes over the cube)",
                           "moments_f64_kernel<2,true,false,0> + moments_f64_kernel<2,true,false,1>", ms, vox * 18 + ny * nx * 24, vox, {"rows_checked": TR, "max_rel_err": err},
                           "2 x (8 B data + 1 B mask) read per voxel, three f64 maps out"))
    # spectral_smooth, 33 taps
    k1 = Gaussian1DKernel(4).array
    ms = event_ms(lambda: ops.spectral_conv_f64(cube, k1, mask=spec, out=out), device)
    err = rel(fetch_rows(out, ny - TR, ny)[:, :, :128], O.spectral_smooth(tile[:, :, :128], inc[:, :, :128], k1), "f64 spectral_smooth")
    recs.append(cfg_record("w_spectral_f64", A64, "float64 spectral_smooth(33 taps), 512x1024x1024 f64 + uint8 mask", "spectral64_ring_kernel<33,true>", ms, vox * 17, vox,
                           {"voxels_checked": nz * TR * 128, "max_rel_err": err}, "8 + 1 read, 8 written per voxel (ring streaming: every input read once; the mask admits infinities, so the runs-of-16 kernel is queued behind the ring kernel's flag and retires at once)"))
    # spatial_smooth, 29 x 29
    k2 = Gaussian2DKernel(8 / 2.3548200450309493).array
    ms = event_ms(lambda: ops.spatial_conv_f64(cube, k2, mask=spec, out=out), device, n=5, warm=1)
    WX, PAD = 160, 14
    sub = np.tile(tile[:2, :, :WX + PAD], (1, 6, 1))                        # 48 rows of the periodic cube: rows 16 .. 32 see no edge
    exp = O.spatial_smooth(sub, np.tile(inc[:2, :, :WX + PAD], (1, 6, 1)), k2)[:, 16:32, PAD:WX]
    got = np.stack([out.planes(z, z + 1).get()[0, 512 // scale:512 // scale + 16, PAD:WX] for z in range(2)])
    err = rel(got, exp, "f64 spatial_smooth")
    recs.append(cfg_record("w_spatial_f64", A64, "float64 spatial_smooth(29x29, outer product), 512x1024x1024 f64 + uint8 mask", "spatial64_ring_kernel<33,true>",
                           ms, vox * 17, vox, {"voxels_checked": int(exp.size), "max_rel_err": err}, "8 + 1 read, 8 written per voxel (one kernel: x pass from a staged row segment, y pass on a register ring; the mask admits infinities, so the two-pass kernels are queued behind its flag and retire at once)"))
    # spectral_interpolate 512 -> 512 channels (shifted grid)
    v = np.arange(nz) * 1.0
    grid = np.linspace(v[0] + 0.25, v[-1] - 0.25, nz)
    lo, t, inv, _, _, fill = ops.lerp_plan(v, grid)
    ms = event_ms(lambda: ops.spectral_lerp_f64(cube, lo, t, inv, fill, out=out), device)
    exp, _ = O.spectral_interpolate(tile[:, :, :128], None, v, grid)
    err = rel(fetch_rows(out, ny - TR, ny)[:, :, :128], exp, "f64 spectral_interpolate", 1e-13)
    recs.append(cfg_record("w_lerp_f64", A64, "float64 spectral_interpolate 512 -> 512 channels, 512x1024x1024 f64", "spectral_lerp64_kernel", ms, vox * 16, vox,
                           {"voxels_checked": int(exp.size), "max_rel_err": err}, "8 read + 8 written per voxel"))
    # median along z
    med = {}

    def median():
        med["m"] = ops.percentile_axis0_f64(cube, 50.0, mask=spec)
    ms = event_ms(median, device, n=5, warm=1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = np.nanmedian(np.where(inc, tile, np.nan), axis=0)
    assert np.array_equal(med["m"].get()[:TR], exp, equal_nan=True), "float64 median(axis=0) differs from np.nanmedian"
    recs.append(cfg_record("w_median_f64", A_SELECT.replace("f32", "f64"), "float64 median(axis=0), 512x1024x1024 f64 + uint8 mask", "select64_reg_kernel<32,16> (keys in registers, 16 lanes per ray = one DPP row)", ms,
                           vox * 9 + ny * nx * 8, vox, {"rows_checked": TR, "vs_np_nanmedian": "bit-identical"}, "8 B data + 1 B mask read per voxel, one f64 map out"))
    # sigma clipping
    keep = {}

    def clip():
        keep["r"] = None
        keep["r"] = ops.sigma_clip_axis0_f64(cube, sigma=3.0, mask=spec)
    ms = event_ms(clip, device, n=3, warm=1)
    got = fetch_rows(keep["r"], 0, TR)
    exp = O.sigma_clip(tile, inc, 3.0, out_dtype=np.float64)
    assert np.array_equal(np.isnan(got), np.isnan(exp)) and np.array_equal(got[~np.isnan(exp)], exp[~np.isnan(exp)]), "float64 sigma clip vs oracle"
    keep.clear()
    recs.append(cfg_record("w_clip_f64", A_CLIP.replace("f32", "f64"), "float64 sigma_clip_spectrally(3), astropy defaults, 512x1024x1024 f64 + uint8 mask", "sigma_clip64_reg_kernel<32,16> (keys in registers, 16 lanes per ray = one DPP row)", ms,
                           vox * 17, vox, {"rows_checked": TR, "clipped_set_and_kept_values": "identical to the oracle"}, "8 + 1 read, 8 written per voxel"))
    for r in recs:
        r["mask_valid_fraction"] = float(inc.mean())
    return recs


def config_records(args, device):
    import gc
    from spectral_cube_amd.device import pool_trim
    fns = {"C3": config_c3, "C4": config_c4, "C5": config_c5}
    which = ("C3", "C4", "C5") if not args.configs_only else tuple(w.strip().upper() for w in args.configs_only.split(","))
    which = tuple(w for w in which if w in fns)             # "--configs-only none": the headline and the next rows alone
    out = {}
    for w in which:
        t0 = time.perf_counter()
        try:
            out[w] = fns[w](device, max(1, args.configs_scale))
        except AssertionError as exc:                       # a failed oracle check is reported, never hidden
            out[w] = {"error": "oracle check failed: %r" % (exc,)}
        gc.collect()
        pool_trim(device)
        print("[bench] %s records in %.1f s" % (w, time.perf_counter() - t0), file=sys.stderr, flush=True)
    return out


# ---- N > 1: strong scaling of the north-star cube -----------------------------------------------
STITCH_EXIT = 4


def require_device_stitch(stitch, local_world, ndev, rank=0):
    """With one GPU per rank the stitch MUST be the RCCL all-gather: a run that fell back to the host rendezvous would put a
    78 ms/step host copy into `value` and still look like a scaling point (profiles/r03_bench_n2_on_1gpu_hostfallback.log).
    Ranks that SHARE a device (a 1-GPU box driven with --gpus 2: RCCL refuses two ranks on one device) may fall back - the
    line says stitch = host-fallback; SPC_BENCH_ALLOW_HOST_STITCH=1 lifts the check."""
    if stitch == "rccl" or local_world > ndev or os.environ.get("SPC_BENCH_ALLOW_HOST_STITCH", "0") == "1":
        return
    print("[bench] rank %d: %d ranks on %d devices and the stitch is %r, not rccl: refusing to benchmark the host fallback "
          "(SPC_BENCH_ALLOW_HOST_STITCH=1 to run it anyway)" % (rank, local_world, ndev, stitch), file=sys.stderr, flush=True)
    sys.exit(STITCH_EXIT)


def rccl_self_check(comm, device, rdv, init_ms, stitch_bytes):
    """before any timing, per rank: a 1 MiB all-gather whose content is checked word by word on the host (rank r sends
    r * 2^20 + i), then the median time of the all-gather at the size the stitch uses, S(N) = 3 maps x rows x NX x 8 B per
    rank.  One line per rank on stderr; a wrong gather exits non-zero."""
    import numpy as np
    from spectral_cube_amd.device import DeviceArray, Event, Stream
    world, rank = rdv.world_size, rdv.rank
    n = (1 << 20) // 8 // world
    send = DeviceArray.from_numpy(np.arange(n, dtype=np.int64) + (rank << 20), device)
    recv = DeviceArray((world, n), np.int64, device)
    st = Stream(device)
    comm.allgather_rows_device(send, recv, st)
    st.synchronize()
    got = recv.get()
    exp = np.arange(n, dtype=np.int64)[None, :] + (np.arange(world, dtype=np.int64)[:, None] << 20)
    ok = bool(np.array_equal(got, exp))
    big_s = DeviceArray((max(stitch_bytes, 8) // 8,), np.float64, device)
    big_r = DeviceArray((world, max(stitch_bytes, 8) // 8), np.float64, device)
    e0, e1, ts = Event(device), Event(device), []
    for i in range(12):
        rdv.barrier()
        e0.record(st)
        comm.allgather_rows_device(big_s, big_r, st)
        e1.record(st)
        e1.synchronize()
        if i >= 2:
            ts.append(e0.elapsed_ms(e1))
    ms = Ms(ts)
    print("[bench] rank %d/%d on device %d: RCCL self-check %s - init %.1f ms, 1 MiB all-gather checksum %s, all-gather of S(N) = "
          "%.2f MiB per rank: median %.3f ms (min %.3f, max %.3f)" % (rank, world, device, "ok" if ok else "FAILED", init_ms,
          "%d" % int(got.sum() % 1000003) if ok else "MISMATCH", stitch_bytes / 2**20, ms, ms.stats["min"], ms.stats["max"]),
          file=sys.stderr, flush=True)
    oks = rdv.allgather_object(ok)
    if not all(oks):
        sys.exit(5)
    return {"init_ms": init_ms, "checksum_ok": True, "allgather_ms_at_stitch_size": float(ms), "stitch_bytes_per_rank": int(stitch_bytes)}


def run_sharded(args, device, rdv):
    import numpy as np
    from spectral_cube_amd import synth
    from spectral_cube_amd.device import DeviceArray, Event, Stream, device_info, synchronize
    from spectral_cube_amd.distributed import HostGatherComm, RcclComm, strip_bounds
    rank, world = rdv.rank, rdv.world_size
    NZ, NY, NX = tuple(args.shape)
    if NY % world:
        raise SystemExit("the %d rows of the cube must divide over %d ranks" % (NY, world))
    y0, y1 = strip_bounds(NY, world, rank)
    rows = y1 - y0
    cube, maskd, tile, tmask = tiled_strip_on_device((NZ, rows, NX), synth.SEEDS["C4"] + 17 * rank, device)
    wl = Workload(cube, maskd, device)

    # RcclComm completes the id broadcast on every rank whether or not rank 0 could make an id (the sequence of
    # rendezvous collectives is the same on all ranks whatever fails); rccl or host-fallback is then decided TOGETHER
    t_init = time.perf_counter()
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
    from spectral_cube_amd import _lib as _spclib
    require_device_stitch(stitch, int(os.environ.get("LOCAL_WORLD_SIZE", world)), _spclib.device_count(), rank)
    self_check = None
    if stitch == "rccl":
        self_check = rccl_self_check(comm, device, rdv, (time.perf_counter() - t_init) * 1e3, 3 * rows * NX * 8)

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

    # (1b) the same ONE call with the stitch hidden inside it (distributed.ChunkedMoments): the rank's rows in 2 - 4 blocks,
    # the all-gather of a block's three maps on a second stream under the kernel of the next block.  Nothing is carried
    # across calls: a call starts when the previous one has completed.  Needs the device all-gather.
    chunked, elapsed_chunked = None, None
    from spectral_cube_amd.distributed import ChunkedMoments
    if stitch == "rccl" and ChunkedMoments.pick_chunks(rows) > 1:
        chunked = ChunkedMoments(cube, maskd, wl.d_cen, 500.0, wl.cref + wl.v[0], comm, workspace=wl.ws)   # blocks of >= 128 rows

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
            # (bit for bit when both launches split z the same way; a strip of more than 2^21 spaxels - one rank forced through
            #  this path - takes four waves per block where its row blocks take eight: the float64 partial sums are then added in
            #  another order, 1e-16 relative)
            for k in range(3):
                a_, b_ = hm[k][g0:g1], full[rank, k][c * chunked.rc:(c + 1) * chunked.rc]
                fin_ = np.isfinite(b_)
                scale_ = float(np.max(np.abs(b_[fin_]))) if fin_.any() else 1.0
                assert np.array_equal(np.isnan(a_), np.isnan(b_)) and (not fin_.any() or float(np.max(np.abs(a_[fin_] - b_[fin_]))) <= 1e-12 * scale_), \
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
    nchunks = chunked.chunks if chunked is not None else 0
    call_form = ("rows in %d blocks, all-gather of a block under the kernel of the next" % nchunks) if best != elapsed else "one launch, then one all-gather"
    head = {"value": total * args.steps / best / 1e6, "unit": "Mvoxel/s", "ms_per_step": best / args.steps * 1e3,
            "roofline": roofline(wl, k_ms_max), "verify": {"per_rank": verifies, "stitched_maps_identical_on_all_ranks": True},
            "verify_compact": {"vs": "oracle, every rank its own rows", "rows_checked": 4 * world,
                               "max_scaled_err": sig(max(max(v["max_scaled_err_m0_m1_m2"]) for v in verifies), 3), "tolerance": 1e-5,
                               "stitched_maps_identical_on_all_ranks": True},
            "config": {"workload": "north star: fixed %dx%dx%d fp32 cube + uint8 mask sharded by row strips over %d GPUs "
                                   "(%d rows each), fused moment0+1+2 per strip + the all-gather stitching the three "
                                   "float64 maps on every rank; ONE call at a time%s" % (
                                       NZ, NY, NX, world, rows,
                                       (" - the rank's rows in %d blocks (block-cyclic ownership), the all-gather of a block's maps "
                                        "under the kernel of the next block, nothing carried across calls" % nchunks) if best != elapsed
                                       else ", kernel then stitch, nothing overlapped"),
                       "input_dtype": "f32 cube + u8 mask, sums carried in f64, f64 maps out", "arithmetic": ARITH_MOMENTS,
                       "mask_valid_fraction": sig(float(np.mean(vfrac)), 4), "stitch": stitch,
                       "sharding": "row strips (nz, %d, nx) of a %dx%dx%d cube" % (rows, NZ, NY, NX),
                       "allgather_bytes_per_rank": 3 * rows * NX * 8,
                       "device": device_info(device)["name"] or device_info(device)["arch"]}}
    detail = {"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "records_file": os.path.basename(args.records_file),
              "headline": head,
              "per_call": {"kernel_ms": k_ms_max, "allgather_ms": g_ms, "latency_ms": best / args.steps * 1e3,
                           "latency_unoverlapped_ms": elapsed / args.steps * 1e3,
                           "latency_chunked_ms": None if elapsed_chunked is None else elapsed_chunked / args.steps * 1e3,
                           "chunks": nchunks, "pipelined_ms_per_step": elapsed_pipe / args.steps * 1e3,
                           "pipelined_mvoxel_per_s": total * args.steps / elapsed_pipe / 1e6},
              "per_call_form": call_form,
              "pipelined_note": "all-gather of step k on its own stream under the kernel of step k+1 (double buffered)",
              "rccl_self_check": self_check, "cpu_baseline": None}
    return detail


# ---- self-launch: python bench.py --gpus N with no launcher -------------------------------------------
def spawn_ranks(n, argv):
    """Start the N ranks of this script (one process per GPU), wait for them, relay rank 0's stdout (its last
    line is the JSON record).  Returns the exit status: 0 only if EVERY rank exited 0.  A rank that dies takes
    the launch down at once - the others would wait for it in the rendezvous until its timeout."""
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    rdv_dir = tempfile.mkdtemp(prefix="spc_rdv_bench_", dir=base)
    import socket
    with socket.socket() as sk:                              # a free port, by convention only (nothing listens on it:
        sk.bind(("127.0.0.1", 0))                            # the ranks meet through files, RCCL through its own id)
        port = sk.getsockname()[1]
    procs, out0 = [], []
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SPC_RDV_DIR=rdv_dir)
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                          stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
        reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout), daemon=True)
        reader.start()
        failed = None
        while failed is None and any(p.poll() is None for p in procs):
            for r, p in enumerate(procs):
                if p.poll() not in (None, 0):
                    failed = (r, p.returncode)
            time.sleep(0.05)
        if failed is None:
            failed = next(((r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0), None)
        if failed is not None:
            for p in procs:                                  # exactly the processes started above
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
        reader.join(timeout=10)
    finally:
        import shutil
        shutil.rmtree(rdv_dir, ignore_errors=True)
    sys.stdout.write("".join(out0))
    sys.stdout.flush()
    if failed is not None:
        print("[bench] rank %d exited with status %s; launch of %d ranks aborted" % (failed[0], failed[1], n), file=sys.stderr)
        return failed[1] if isinstance(failed[1], int) and failed[1] > 0 else 1
    return 0


def dry_run(args):
    """SPC_BENCH_DRYRUN=1: everything of a multi-rank launch except the GPU work (the CPU test of the spawner):
    rendezvous, a few collectives, rank 0's JSON line; SPC_BENCH_DRYRUN_FAIL_RANK=r makes rank r exit 3."""
    from spectral_cube_amd.rendezvous import FileRendezvous, SingleProcess
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    if os.environ.get("SPC_BENCH_DRYRUN_FAIL_RANK", "") == str(rank):
        sys.exit(3)
    rdv = FileRendezvous.from_env(timeout=60) if world > 1 else SingleProcess()
    try:
        # SPC_BENCH_DRYRUN_STITCH / _DEVICES: the stitch decision of run_sharded without a GPU (host-fallback with a device
        # per rank must end the launch with a non-zero status)
        if "SPC_BENCH_DRYRUN_STITCH" in os.environ:
            flags = rdv.allgather_object(os.environ["SPC_BENCH_DRYRUN_STITCH"] if rank == world - 1 else "rccl")
            stitch = "rccl" if all(f == "rccl" for f in flags) else "host-fallback"
            require_device_stitch(stitch, int(os.environ.get("LOCAL_WORLD_SIZE", world)),
                                  int(os.environ.get("SPC_BENCH_DRYRUN_DEVICES", world)), rank)
        token = rdv.bcast_bytes(b"id-from-rank-0" if rank == 0 else None)
        ranks = rdv.allgather_object((rank, int(os.environ.get("LOCAL_RANK", -1)), os.getpid()))
        rdv.barrier()
    finally:
        rdv.close()
    if rank == 0:
        print("noise before the record")
        # the line a real run of this world size prints, from canned records (the CPU test of the line's size and shape)
        detail = canned_detail(world, args)
        line = emit(detail, os.environ.get("SPC_BENCH_DRYRUN_RECORDS", os.devnull))
        line["dryrun"] = {"gpus_arg": args.gpus, "ranks": ranks, "bcast": token.decode()}
        print(json.dumps(line), flush=True)


CANNED_KEYS = ("f1_stats", "f4_median", "f4_clip", "f4_median_dense", "f4_clip_dense", "c3_fused", "c3_mat", "c3_fused_mask", "c3_fused_mask_f32acc",
               "c3_mat_mask", "c3_mat_f32acc", "c4_mat", "c4_mat_mask_ring", "c4_mat_mask", "c4_pipe_mask", "c4_fused_mask", "c4_fused012_mask",
               "c4_mat_sigmask_ring", "c4_fused_sigmask", "c4_fused", "c5_lerp", "c5_reproject", "c5_one_pass", "w_stats_f64",
               "w_spectral_f64", "w_spatial_f64", "w_median_f64", "w_clip_f64", "w_moments_f64", "w_lerp_f64")


def canned_detail(world, args):
    """a full record with the shape, key set and value widths of a real run (numbers of round 5), for the CPU tests of
    compact_line / emit and the multi-rank dry run"""
    class W:
        alg_bytes = 86000009216 // world
        shape = (4096, 2048 // world, 2048)
    k = Ms([13.706123456 / world * f for f in (0.97, 1.0, 1.01, 1.013, 0.99)])
    ver = {"rows_checked": 4, "max_scaled_err_m0_m1_m2": [1.1234567e-16, 2.2345678e-13, 3.3456789e-12], "nan_pattern": "identical"}
    head = {"value": 1253421.987654 * world * 0.9, "unit": "Mvoxel/s", "ms_per_step": 13.7123456 / world / 0.9,
            "mask_valid_fraction": 0.0512345678, "roofline": roofline(W, k, 86001234567.0 / world, PMC_FILE + "#ns"), "verify": ver,
            "verify_compact": {"vs": "oracle (numpy float64 restatement of the reference)", "rows_checked": 4, "max_scaled_err": 3.35e-12,
                               "tolerance": 1e-5, "nan_pattern": "identical"},
            "config": {"workload": "north star: fixed 4096x2048x2048 fp32 cube + uint8 mask sharded by row strips over %d GPUs (%d rows each), "
                                   "fused moment0+1+2 per strip + the all-gather stitching the three float64 maps on every rank; ONE call at a "
                                   "time - the rank's rows in 4 blocks (block-cyclic ownership), the all-gather of a block's maps under the "
                                   "kernel of the next block, nothing carried across calls" % (world, 2048 // world),
                       "input_dtype": "f32 cube + u8 mask, sums carried in f64, f64 maps out", "arithmetic": ARITH_MOMENTS,
                       "mask_valid_fraction": 0.05123, "stitch": "rccl", "sharding": "row strips (nz, %d, nx) of a 4096x2048x2048 cube" % (2048 // world),
                       "allgather_bytes_per_rank": 3 * (2048 // world) * 2048 * 8, "data_note": "device-tiled synthetic data: one seeded 16-row host tile repeated along y",
                       "device": "AMD Instinct MI355X"}}
    detail = {"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "records_file": "bench_records.json", "headline": head}
    if world > 1:
        detail["per_call"] = {"kernel_ms": 1.634567, "allgather_ms": 0.5512345, "latency_ms": 1.9123456, "latency_unoverlapped_ms": 2.2123456,
                              "latency_chunked_ms": 1.9123456, "chunks": 2, "pipelined_ms_per_step": 1.71234567, "pipelined_mvoxel_per_s": 1.0034567e7}
        detail["cpu_baseline"] = None
        return detail
    head["strip_terms"] = [{"n_gpus_modelled": n, "kernel_ms": 13.706 / n * 1.0123, "blocks_ms": 13.706 / n * 1.0634, "blocks": 4 if n < 8 else 2}
                           for n in (2, 4, 8)]
    W.alg_bytes, W.shape = 5393874944, (1024, 1024, 1024)
    detail["configs1"] = {"workload": "configs[1]: 1024x1024x1024 fp32 + uint8 mask, fused moment0+1+2", "value": 1.2234567e6, "unit": "Mvoxel/s",
                          "mask_valid_fraction": 0.051, "roofline": roofline(W, Ms([0.8775123, 0.88, 0.87]), 5394218316.8, PMC_FILE + "#c2"), "verify": ver}
    recs = [cfg_record(key, A_SPAT_SPLIT + "; f32 channel-chunk sums, f64 map", "record %s: " % key + "x" * 150, "kernel_name<1,2,3> " + "y" * 100,
                       Ms([39.8123456 + i, 40.0 + i, 39.7 + i]), 85899345920 + 33554432, 4096 * 2048 * 2048,
                       {"max_scaled_err": 1.0234e-6, "voxels_checked": 92160}, "4 + 1 read + 8 B/spaxel out", mask_valid_fraction=0.8)
            for i, key in enumerate(CANNED_KEYS)]
    for r in recs:
        r["traffic_over_algorithmic"] = 1.5312345
    detail["next_rows"] = {r["key"]: r for r in recs[:5]}
    detail["configs"] = {"C3": recs[5:11], "C4": recs[11:20], "C5": recs[20:23]}
    detail["wide"] = recs[23:]
    detail["cpu_baseline"] = {"value": 130.912, "unit": "Mvoxel/s", "cores": 64, "kind": "port", "single_thread_value": 8.71234, "seconds": 21.31,
                              "sample": "128 strips of 4096x1x2048 voxels of the headline cube (moment 0,1,2 = three reference passes each), numpy float64 oracle, ThreadPool(64)",
                              "reference_anchor": "profiles/r01_reference_cpu_buildbox.txt (8 cores, 256^3): Dask class 4.6 Mvoxel/s, NumPy class 6.8, this port on one thread 8.7"}
    return detail


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, argv))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        args.gpus = world                                   # under a launcher its world size wins
    if os.environ.get("SPC_BENCH_DRYRUN", "0") == "1":
        return dry_run(args)
    global PMC_ON
    PMC_ON = (tuple(args.shape) == NORTH_STAR and tuple(args.configs1_shape) == (1024, 1024, 1024) and args.configs_scale == 1)
    from spectral_cube_amd import _lib
    from spectral_cube_amd.rendezvous import FileRendezvous, SingleProcess
    _lib.require_gpu()
    device = local_rank % _lib.device_count()
    # SPC_BENCH_FORCE_DIST=1: take the sharded path with one rank (rendezvous + RCCL init + stitch on a 1-GPU box)
    sharded = world > 1 or os.environ.get("SPC_BENCH_FORCE_DIST", "0") == "1"
    if sharded:
        rdv = FileRendezvous.from_env() if world > 1 else SingleProcess()
        try:
            detail = run_sharded(args, device, rdv)
        finally:
            rdv.close()                                     # also after a failure: no stale files for a restarted launch
    else:
        detail = run_single(args, device)
        detail["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(tuple(args.shape), args.cpu_seconds)
    line = emit(detail, args.records_file) if rank == 0 else None
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
