"""CPU, world_size 2 over gloo: the N>1 path (row-strip sharding + the single
all-gather stitch).  No HIP compute can run here, so each rank's strip maps
come from the oracle - what is under test is strip_bounds, padding of a short
last strip and the gather order."""
import os
import socket
import subprocess
import sys

import numpy as np

from conftest import REPO

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "oracle"))
import torch.distributed as dist
import oracle_np as O
from spectral_cube_amd import synth
from spectral_cube_amd.distributed import HostGatherComm, strip_bounds, sharded_statistics
dist.init_process_group("gloo", init_method="env://")
rank, ws = dist.get_rank(), dist.get_world_size()
shape = (24, 9, 5)                     # 9 rows over 2 ranks: strips of 5 and 4 rows
d = synth.gaussian_line_cube(shape, 77)
inc = synth.boolean_mask(d, 3).astype(bool)
cen = np.arange(24.0) * 2.0
y0, y1 = strip_bounds(shape[1], ws, rank)
comm = HostGatherComm()
full = [comm.allgather_rows(O.moment(d[:, y0:y1], inc[:, y0:y1], o, cen, 2.0), shape[1]) for o in range(3)]
ids = comm.allgather_rows(O.argmax(d[:, y0:y1], inc[:, y0:y1]), shape[1])
st = sharded_statistics(O.statistics(d[:, y0:y1], inc[:, y0:y1]))
comm.barrier()
es = O.statistics(d, inc)
assert st["npts"] == es["npts"] and st["min"] == es["min"] and st["max"] == es["max"]
for k in ("sum", "sumsq", "mean", "sigma", "rms"):
    assert abs(st[k] - es[k]) <= 1e-12 * abs(es[k]), k
if rank == 0:
    for o in range(3):
        exp = O.moment(d, inc, o, cen, 2.0)
        assert full[o].shape == exp.shape
        assert np.array_equal(np.isnan(full[o]), np.isnan(exp))
        assert np.array_equal(full[o][~np.isnan(exp)], exp[~np.isnan(exp)])
    assert np.array_equal(ids, O.argmax(d, inc))
    print("DIST_OK")
dist.destroy_process_group()
'''


def test_two_rank_gloo_stitch(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, str(script), REPO], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "DIST_OK" in outs[0]
