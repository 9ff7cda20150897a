"""Scratch/report: the dask seam end to end (PCIe inclusive) on the GPU box - run under the interpreter that has
dask (/opt/conda/bin/python3.9 -B tools/bench_dask.py [n]).  A host-resident n^3 float32 cube goes through
dask.array.map_blocks with the chunk functions of spectral_cube_amd.dask_adapter (chunking (-1, cy, cx) as
apply_function_parallel_spectral rechunks, dask_spectral_cube.py:551,618), `threads` scheduler with 1 / 4 / 8
workers; next to it the same graph with the numpy chunk arithmetic of the reference's moment()
(dask_spectral_cube.py:1083-1104, restated in oracle/oracle_np.py) - the Dask class itself only imports in the build
container (profiles/r01_reference_cpu_buildbox.txt: 4.6 Mvoxel/s for moment 0+1+2 at 256^3)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "oracle"))
import numpy as np
import dask, dask.array as da
import oracle_np as O
from spectral_cube_amd.dask_adapter import MomentChunk, Moments012Chunk, SpectralSmoothChunk
from spectral_cube_amd.kernels import Gaussian1DKernel
from spectral_cube_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
shape = (n, n, n)
d = np.tile(synth.gaussian_line_cube((n, 16, n), 5, chunk_rows=16), (1, n // 16, 1))
vox = d.size
cen = np.arange(n, dtype=np.float64) * 500.0
cy = cx = max(64, n // 4)                              # 16 chunks of (n, n/4, n/4): 32 MiB at 512^3, 256 MiB at 1024^3
arr = da.from_array(d, chunks=(-1, cy, cx))
k1 = Gaussian1DKernel(4).array


def numpy_m012(chunk):
    inc = np.isfinite(chunk)
    return np.stack(O.moments012(chunk, inc, cen, 500.0, 0.0))


def run(label, graph, workers, reps=3):
    ts = []
    with dask.config.set(scheduler="threads", num_workers=workers):
        for _ in range(reps):
            t0 = time.perf_counter(); out = graph.compute(); ts.append(time.perf_counter() - t0)
    best = min(ts)
    print("%-64s workers=%d  %8.1f ms  %8.0f Mvoxel/s  %5.1f GB/s in" % (label, workers, best * 1e3, vox / best / 1e6, vox * 4 / best / 1e9), flush=True)
    return out


print("cube %s float32 = %.2f GiB, chunks (-1, %d, %d)" % (shape, d.nbytes / 2**30, cy, cx), flush=True)
ref = None
for w in (1, 4, 8):
    g = da.map_blocks(Moments012Chunk(cen, 500.0), arr, dtype=np.float64, drop_axis=[0], new_axis=[0], chunks=((3,), arr.chunks[1], arr.chunks[2]))
    ref = run("GPU Moments012Chunk (one staging, three maps)", g, w)
for w in (1, 8):
    gs = [da.map_blocks(MomentChunk(o, cen, 500.0), arr, dtype=np.float64, drop_axis=[0], chunks=(arr.chunks[1], arr.chunks[2])) for o in (0, 1, 2)]
    run("GPU MomentChunk x 3 orders (chunk staged three times)", da.stack(gs), w)
for w in (1, 4, 8):
    g = da.map_blocks(SpectralSmoothChunk(k1), arr, dtype=arr.dtype)
    run("GPU SpectralSmoothChunk 33 taps (cube -> cube)", g, w)
if n <= 512:
    for w in (1, 8, 32):
        g = da.map_blocks(numpy_m012, arr, dtype=np.float64, drop_axis=[0], new_axis=[0], chunks=((3,), arr.chunks[1], arr.chunks[2]))
        got = run("numpy chunk arithmetic of the reference's moment 0+1+2", g, w, reps=1)
    for a, b, sc in zip(ref, got, (np.nanmax(np.abs(got[0])), 500.0 * n, np.nanmax(np.abs(got[2])))):
        assert np.nanmax(np.abs(a - b)) <= 1e-5 * sc
    print("GPU maps == numpy maps (1e-5)")
