"""CPU, world_size 2: the N>1 path (row-strip sharding + the single all-gather stitch) over
both host transports - torch.distributed / gloo (a transport class local to this test: the
product package imports no torch) and the package's own file rendezvous.  No HIP compute can run
here, so each rank's strip maps come from the oracle - what is under test is strip_bounds,
padding of a short last strip, the gather order, the statistics combine and the rendezvous."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO

WORKER = r'''
import os, sys, pickle
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "oracle"))
import oracle_np as O
from spectral_cube_amd import synth
from spectral_cube_amd.distributed import HostGatherComm, strip_bounds, sharded_statistics, sharded_percentile
from spectral_cube_amd.rendezvous import FileRendezvous

kind = sys.argv[2]
if kind == "gloo":
    import torch, torch.distributed as dist
    dist.init_process_group("gloo", init_method="env://")

    class GlooTransport:                      # the rendezvous protocol over torch.distributed (test only)
        rank, world_size = dist.get_rank(), dist.get_world_size()
        def allgather_bytes(self, payload):
            out = [None] * self.world_size
            dist.all_gather_object(out, bytes(payload))
            return out
        def bcast_bytes(self, payload=None):
            obj = [payload]
            dist.broadcast_object_list(obj, src=0)
            return obj[0]
        def allgather_object(self, obj):
            return [pickle.loads(b) for b in self.allgather_bytes(pickle.dumps(obj))]
        def barrier(self):
            dist.barrier()
        def close(self):
            dist.destroy_process_group()
    tr = GlooTransport()
else:
    tr = FileRendezvous.from_env(timeout=120)
rank, ws = tr.rank, tr.world_size
assert tr.bcast_bytes(b"id-from-rank-0" if rank == 0 else None) == b"id-from-rank-0"
shape = (24, 9, 5)                     # 9 rows over 2 ranks: strips of 5 and 4 rows
d = synth.gaussian_line_cube(shape, 77)
inc = synth.boolean_mask(d, 3).astype(bool)
cen = np.arange(24.0) * 2.0
y0, y1 = strip_bounds(shape[1], ws, rank)
comm = HostGatherComm(tr)
full = [comm.allgather_rows(O.moment(d[:, y0:y1], inc[:, y0:y1], o, cen, 2.0), shape[1]) for o in range(3)]
ids = comm.allgather_rows(O.argmax(d[:, y0:y1], inc[:, y0:y1]), shape[1])
st = sharded_statistics(O.statistics(d[:, y0:y1], inc[:, y0:y1]), tr)
comm.barrier()
es = O.statistics(d, inc)
assert st["npts"] == es["npts"] and st["min"] == es["min"] and st["max"] == es["max"]
for k in ("sum", "sumsq", "mean", "sigma", "rms"):
    assert abs(st[k] - es[k]) <= 1e-12 * abs(es[k]), k
for o in range(3):
    exp = O.moment(d, inc, o, cen, 2.0)
    assert full[o].shape == exp.shape
    assert np.array_equal(np.isnan(full[o]), np.isnan(exp))
    assert np.array_equal(full[o][~np.isnan(exp)], exp[~np.isnan(exp)])
assert np.array_equal(ids, O.argmax(d, inc))
# whole-cube order statistics of the sharded cube: the per-rank passes restated in numpy (the device passes are
# spc_key_histogram_f32), the exchange + walk are the product code
def _keys(a, inc_, center):
    v = a[inc_ & np.isfinite(a) | (inc_ & np.isinf(a))].astype(np.float32)
    if center is not None:
        v = np.abs(v - np.float32(center))
    u = v.view(np.uint32)
    return np.where(u & 0x80000000, ~u, u | np.uint32(0x80000000)).astype(np.uint32)
def _hist(strip, prefix, pmask, shift, mask=None, center=None):
    k = _keys(strip, mask, center)
    k = k[(k & np.uint32(pmask)) == np.uint32(prefix)]
    return np.bincount((k >> np.uint32(shift)) & 0xff, minlength=256).astype(np.uint64)
def _next(strip, prefix, mask=None, center=None):
    k = _keys(strip, mask, center)
    k = k[k > np.uint32(prefix)]
    return int(k.min()) if k.size else 0xffffffff
def _unkey(key):
    u = np.uint32(key)
    return float(np.array([u & 0x7fffffff if u & 0x80000000 else ~u], np.uint32).view(np.float32)[0])
fz = np.where(inc, d, np.nan).astype(np.float32)
for q in (0.0, 12.5, 50.0, 99.0, 100.0):
    got = sharded_percentile(d[:, y0:y1], q, tr, mask=inc[:, y0:y1], passes=(_hist, _next, _unkey))
    exp = float(np.nanpercentile(fz.astype(np.float64), q))
    assert abs(got - exp) <= 2e-7 * max(1.0, abs(exp)), (q, got, exp)
med = float(np.nanmedian(fz))
mad = sharded_percentile(d[:, y0:y1], 50.0, tr, mask=inc[:, y0:y1], center=med, passes=(_hist, _next, _unkey))
assert mad == float(np.nanmedian(np.abs(fz - np.float32(med))))
assert np.isnan(sharded_percentile(d[:, y0:y1], 50.0, tr, mask=np.zeros_like(inc[:, y0:y1]), passes=(_hist, _next, _unkey)))
for i in range(20):                    # many small collectives back to back (file retirement)
    got = tr.allgather_object((rank, i))
    assert got == [(r, i) for r in range(ws)]
tr.close()
if rank == 0:
    print("DIST_OK")
'''


@pytest.mark.parametrize("kind", ["gloo", "file"])
def test_two_rank_stitch(tmp_path, kind):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), LOCAL_RANK=str(rank), SPC_RDV_DIR=str(tmp_path / "rdv"))
        procs.append(subprocess.Popen([sys.executable, str(script), REPO, kind], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "DIST_OK" in outs[0]
    if kind == "file":
        assert not (tmp_path / "rdv").exists(), "rank 0 removes the rendezvous directory"


def test_file_rendezvous_default_directory_is_per_launch():
    from spectral_cube_amd.rendezvous import FileRendezvous
    a = FileRendezvous.from_env({"RANK": "0", "WORLD_SIZE": "1", "MASTER_PORT": "29512"})
    b = FileRendezvous.from_env({"RANK": "0", "WORLD_SIZE": "1", "MASTER_PORT": "29513"})
    try:
        assert a.path != b.path and str(os.getppid()) in a.path
        assert a.allgather_object({"x": 1}) == [{"x": 1}]
    finally:
        a.close(); b.close()
    assert not os.path.exists(a.path)


def test_file_rendezvous_ignores_what_an_earlier_launch_left_behind(tmp_path):
    """a crashed launch leaves its files in a directory that the next launch reuses (fixed
    SPC_RDV_DIR, torchrun restart).  The handshake gives every launch a fresh session id carried by every file
    name, so neither stale collective payloads nor stale handshake files are ever read."""
    import pickle
    import threading
    from spectral_cube_amd.rendezvous import FileRendezvous
    d = tmp_path / "rdv"
    d.mkdir()
    # debris of an earlier launch: old-format payloads, handshake files of a dead session, same-format payloads
    (d / "00000000.1").write_bytes(b"STALE-RCCL-ID")
    (d / "hello.1").write_bytes(b"deadbeefdeadbeef")
    (d / "hello.0").write_bytes(b"0123456701234567")
    (d / "session").write_bytes(pickle.dumps(("oldsession000000", [b"0123456701234567", b"deadbeefdeadbeef"])))
    (d / "ack.1").write_bytes(b"oldsession000000")
    (d / "go").write_bytes(b"oldsession000000")
    (d / "oldsession000000.00000000.1").write_bytes(b"STALE-RCCL-ID")
    (d / "oldsession000000.00000000.0").write_bytes(b"STALE-RCCL-ID")
    res, errs = {}, []

    def rank(r, delay):
        try:
            import time
            time.sleep(delay)                     # rank 1 arrives late: rank 0 first sees only the stale hello.1
            tr = FileRendezvous(str(d), r, 2, timeout=30)
            res[r] = (tr.bcast_bytes(b"fresh-id" if r == 0 else None), tr.allgather_object(("rank", r)))
            tr.close()
        except Exception as exc:                  # pragma: no cover
            errs.append(exc)

    ts = [threading.Thread(target=rank, args=(0, 0.0)), threading.Thread(target=rank, args=(1, 0.3))]
    [t.start() for t in ts]
    [t.join(60) for t in ts]
    assert not errs, errs
    for r in (0, 1):
        assert res[r] == (b"fresh-id", [("rank", 0), ("rank", 1)])
    assert not d.exists()


def test_rccl_comm_init_keeps_ranks_in_step_when_rank0_cannot_make_an_id(tmp_path, monkeypatch):
    """a failure on rank 0 BEFORE the id broadcast must not desynchronise the rendezvous sequence:
    the broadcast is completed with an empty id and every rank raises the same error."""
    from spectral_cube_amd import _lib
    from spectral_cube_amd.distributed import RcclComm

    class Tr:
        rank, world_size = 0, 2
        sent = []

        def bcast_bytes(self, payload=None):
            Tr.sent.append(payload)
            return payload

    def boom(name, *a):
        raise _lib.HipLibraryError("no RCCL here")
    monkeypatch.setattr(_lib, "call", boom)
    with pytest.raises(_lib.HipLibraryError, match="RCCL id"):
        RcclComm(0, Tr())
    assert Tr.sent == [b""], "the broadcast happened, with the sentinel"


def _bench(args, **env):
    e = dict(os.environ, **env)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=120)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` with no launcher starts N ranks itself (RANK / LOCAL_RANK /
    WORLD_SIZE in their environment), the ranks meet through the file rendezvous, rank 0's JSON record is the LAST
    line of the parent's stdout, exit status 0.  (SPC_BENCH_DRYRUN=1: everything but the GPU work.)"""
    import json
    p = _bench(["--gpus", "4", "--steps", "3"], SPC_BENCH_DRYRUN="1")
    assert p.returncode == 0, p.stderr
    last = p.stdout.strip().splitlines()[-1]
    rec = json.loads(last)
    dr = rec["dryrun"]
    assert rec["n_gpus"] == 4 and rec["steps"] == 3 and dr["bcast"] == "id-from-rank-0" and dr["gpus_arg"] == 4
    assert [r[:2] for r in dr["ranks"]] == [[i, i] for i in range(4)]
    assert len({r[2] for r in dr["ranks"]}) == 4, "one process per rank"
    _check_line(rec, last)


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline", "cpu_baseline")
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "algorithmic_bytes",
                 "traffic_over_algorithmic", "traffic_source", "arithmetic")


def _check_line(rec, text):
    """round-5 verdict, item 1: the final stdout line stays small enough for the driver to parse (it keeps ~8 KB; the
    round-5 line was 31.6 KB and BENCH_r05.parsed came out null) and carries the contract's keys at the top level"""
    assert len(text) < 8000, len(text)
    for k in CONTRACT_KEYS:
        assert k in rec, k
    for k in ROOFLINE_KEYS:
        assert k in rec["roofline"], k
    assert "4096x2048x2048" in rec["config"]["workload"], "the same cube at every N"
    assert rec["scaling"] == "strong" and rec["unit"] == "Mvoxel/s" and rec["roofline"]["bound"] == "hbm"
    assert abs(rec["roofline"]["frac"] - rec["roofline"]["achieved"] / rec["roofline"]["peak"]) < 1e-3


def test_bench_line_is_small_and_round_trips(tmp_path):
    """the compact line from canned full records (every record key of a real run, names and kernels at their real
    lengths): < 8000 bytes, json round trip, every record reachable under its short key; the full records land in the
    records file"""
    import json
    sys.path.insert(0, REPO)
    import bench
    args = bench.parse([])
    for world in (1, 2, 8):
        detail = bench.canned_detail(world, args)
        rf = tmp_path / ("records_%d.json" % world)
        line = bench.emit(detail, str(rf))
        text = json.dumps(line)
        assert json.loads(text) == line
        _check_line(line, text)
        full = json.loads(rf.read_text())
        assert full["line"] == line and full["headline"]["roofline"]["kernel_ms_stats"]["n"] == 5
        if world == 1:
            assert set(line["records"]) == set(bench.CANNED_KEYS)
            assert all(len(v) == 3 for v in line["records"].values())
            assert line["cpu_baseline"]["cores"] == 64 and line["roofline"]["configs1"]["frac"] > 0
            assert set(line["roofline"]["strip_kernel_ms_at_n"]) == {"2", "4", "8"}
    # a line that would not fit sheds its rows, never its contract keys
    detail = bench.canned_detail(1, args)
    detail["wide"] = [dict(detail["wide"][0], key="k%03d_%s" % (i, "x" * 40)) for i in range(200)]
    line = bench.compact_line(detail)
    assert len(json.dumps(line)) <= bench.LINE_LIMIT and "dropped" in line["records"] and "roofline" in line


def test_bench_eight_rank_dry_run_line():
    """an 8-rank launch of the spawner (no GPU work): rank 0's last stdout line satisfies the same size bound"""
    import json
    p = _bench(["--gpus", "8", "--steps", "2"], SPC_BENCH_DRYRUN="1")
    assert p.returncode == 0, p.stderr
    last = p.stdout.strip().splitlines()[-1]
    rec = json.loads(last)
    assert rec["n_gpus"] == 8 and len(rec["dryrun"]["ranks"]) == 8 and "per_call" in rec
    _check_line(rec, last)


def test_chunked_moments_picks_blocks_from_the_strip_height():
    """round-5 verdict, item 5: 64-row blocks cost the headline kernel 19 %, 128-row ones 6 %: blocks of >= 128 rows"""
    from spectral_cube_amd.distributed import ChunkedMoments
    assert [ChunkedMoments.pick_chunks(2048 // n) for n in (1, 2, 4, 8, 16)] == [4, 4, 4, 2, 1]
    assert ChunkedMoments.pick_chunks(384) == 3 and ChunkedMoments.pick_chunks(130) == 1


def test_bench_launch_fails_when_a_rank_fails():
    import time
    t0 = time.time()
    p = _bench(["--gpus", "3"], SPC_BENCH_DRYRUN="1", SPC_BENCH_DRYRUN_FAIL_RANK="2")
    assert p.returncode == 3 and "rank 2 exited" in p.stderr
    assert time.time() - t0 < 30, "the surviving ranks are stopped, not left to the rendezvous timeout"


def test_bench_without_gpu_fails_loudly_not_silently():
    p = _bench(["--gpus", "1", "--steps", "1", "--no-configs1", "--no-cpu-baseline"])
    assert p.returncode != 0 and "no HIP device" in p.stderr and not p.stdout.strip()


def test_bench_refuses_to_time_the_host_fallback_when_every_rank_has_a_device():
    """RCCL failing on ONE rank of a launch with a GPU per rank makes every rank fall back to the
    host stitch together - and the launch must then END with a non-zero status instead of printing a 78 ms/step scaling
    point.  Ranks sharing a device (--gpus 2 on a 1-GPU box) may fall back; the override is explicit."""
    p = _bench(["--gpus", "2"], SPC_BENCH_DRYRUN="1", SPC_BENCH_DRYRUN_STITCH="host-fallback", SPC_BENCH_DRYRUN_DEVICES="2")
    assert p.returncode == 4 and "refusing to benchmark the host fallback" in p.stderr and not p.stdout.strip().endswith("}")
    p = _bench(["--gpus", "2"], SPC_BENCH_DRYRUN="1", SPC_BENCH_DRYRUN_STITCH="host-fallback", SPC_BENCH_DRYRUN_DEVICES="1")
    assert p.returncode == 0, p.stderr                        # two ranks on one device: the fallback is allowed (and reported)
    p = _bench(["--gpus", "2"], SPC_BENCH_DRYRUN="1", SPC_BENCH_DRYRUN_STITCH="host-fallback", SPC_BENCH_DRYRUN_DEVICES="2",
               SPC_BENCH_ALLOW_HOST_STITCH="1")
    assert p.returncode == 0, p.stderr
    p = _bench(["--gpus", "2"], SPC_BENCH_DRYRUN="1", SPC_BENCH_DRYRUN_STITCH="rccl", SPC_BENCH_DRYRUN_DEVICES="2")
    assert p.returncode == 0, p.stderr
