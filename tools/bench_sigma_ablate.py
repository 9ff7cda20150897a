import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np
from spectral_cube_amd import ops, synth
from spectral_cube_amd.device import DeviceArray, Event, synchronize
from test_gpu_fullsize import _replicate_rows
shape = (1024, 1024, 1024)
tile = synth.gaussian_line_cube((shape[0], 8, shape[2]), 2001, chunk_rows=8)
cube = DeviceArray(shape, np.float32); _replicate_rows(cube, tile, 4)
def timeit(fn, n=3):
    fn(); synchronize(); e0, e1 = Event(), Event(); e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_ms(e1) / n
os.environ["SPC_SIGMA_BT"] = "256"
off = DeviceArray(shape, np.float32); _replicate_rows(off, tile + np.float32(10.0), 4)
for name, c in (("median near 0", cube), ("data + 10", off)):
  print(name)
  for g, what in (("16", "normal"), ("1000", "iterations >= 2 without a descent"), ("1001", "iterations >= 2 with a full descent (no resume)")):
    os.environ["SPC_XCD_GROUP"] = g
    print("  ", what, " ".join("maxiters=%s: %.2f ms" % (it, timeit(lambda: ops.sigma_clip_axis0(c, sigma=3.0, maxiters=it))) for it in (1, 2, 3, 5)), flush=True)
