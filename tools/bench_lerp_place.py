"""Scratch: is the C5 spectral_interpolate time sensitive to where the output sits relative to the input? (r02: 4.66 ms, r03 / r04: 5.4 - 5.5 ms
for the same kernel and bytes)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from spectral_cube_amd import ops, synth
from spectral_cube_amd.device import DeviceArray, Event
from test_gpu_fullsize import _replicate_rows
shape = (2048, 1024, 1024)
tile = synth.gaussian_line_cube((shape[0], 2, shape[2]), 2004, chunk_rows=2)
v = synth.spectral_axis(shape[0]); grid = np.linspace(v[0], v[-1], 4096)
lo, t, inv, _, _, fill = ops.lerp_plan(v, grid)
def run(label, cube, out):
    ts = []
    for i in range(7):
        e0, e1 = Event(), Event()
        e0.record(); ops.spectral_lerp(cube, lo, t, inv, fill, out=out); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_ms(e1))
    print("%-50s median %.3f ms  min %.3f  (cube 0x%x out 0x%x)" % (label, np.median(ts[1:]), min(ts), cube.ptr, out.ptr), flush=True)
cube = DeviceArray(shape, np.float32); _replicate_rows(cube, tile, 4)
out = DeviceArray((4096,) + shape[1:], np.float32)
run("as allocated", cube, out)
nb = 4096 * 1024 * 1024 * 4
for off in (4096, 65536, 1 << 20, (1 << 20) + 4096 * 17, (1 << 21) + 12288):
    big = DeviceArray((nb + (4 << 20),), np.uint8)
    o2 = DeviceArray((4096,) + shape[1:], np.float32, ptr=big.ptr + off, owner=big)
    run("output shifted by %d bytes" % off, cube, o2)
    del o2, big
