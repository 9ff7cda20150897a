"""Time the REAL reference on the build container's CPU cores next to the numpy oracle (BASELINE.md section 3,
item 1): anchors the speed of the oracle - which is what bench.py can time on the GPU box - to the speed of the
reference itself.  TEST INFRASTRUCTURE ONLY; build container only (needs /root/reference):

    /opt/conda/bin/python3.9 -B oracle/time_reference.py > profiles/r01_reference_cpu_buildbox.txt
"""
import os
import sys
import time
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "ref_env"))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from bootstrap import load_reference  # noqa: E402

load_reference()
warnings.simplefilter("ignore")
from astropy import convolution  # noqa: E402
from astropy.io import fits  # noqa: E402
from spectral_cube import SpectralCube  # noqa: E402

import oracle_np as O  # noqa: E402
from gen_golden import c1_header  # noqa: E402  (header only; no case is run on import)
from spectral_cube_amd import synth  # noqa: E402


def best(fn, n=3):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


def main():
    print("# real reference vs numpy oracle on the build container: %d cores, numpy %s" % (os.cpu_count(), np.__version__))
    print("# op | shape | class / scheduler | seconds | Mvoxel/s")
    for shape in ((128, 64, 64), (256, 256, 256)):
        data = synth.gaussian_line_cube(shape, 1234)
        med = float(np.nanmedian(data))
        hdu = fits.PrimaryHDU(data=data, header=c1_header(*shape))
        vox = float(np.prod(shape))
        for label, kw, sched in (("numpy class how=cube", dict(use_dask=False), None),
                                 ("dask class synchronous", dict(use_dask=True), ("synchronous", {})),
                                 ("dask class threads x8", dict(use_dask=True), ("threads", {"num_workers": 8}))):
            sc = SpectralCube.read(hdu, **kw)
            sc.allow_huge_operations = True
            sc = sc.with_mask(sc > med * sc.unit)
            if sched is not None:
                sc.use_dask_scheduler(sched[0], **sched[1])

            def run():
                for order in (0, 1, 2):
                    m = sc.moment(order=order) if kw["use_dask"] else sc.moment(order=order, how="cube")
                    np.asarray(m)
            t = best(run)
            print("moment 0+1+2 | %s | %s | %.4f | %.1f" % ("x".join(map(str, shape)), label, t, vox / t / 1e6))
        include = data > med
        cen = np.arange(shape[0]) * 500.0
        t = best(lambda: O.moments012(data, include, cen, 500.0, 0.0))
        print("moment 0+1+2 | %s | numpy oracle (1 thread) | %.4f | %.1f" % ("x".join(map(str, shape)), t, vox / t / 1e6))
    shape = (128, 128, 128)
    data = synth.gaussian_line_cube(shape, 2002)
    hdu = fits.PrimaryHDU(data=data, header=c1_header(*shape))
    vox = float(np.prod(shape))
    k1 = convolution.Gaussian1DKernel(4)
    k2 = convolution.Gaussian2DKernel(8 / 2.35482)
    for label, kw in (("numpy class", dict(use_dask=False)), ("dask class synchronous", dict(use_dask=True))):
        sc = SpectralCube.read(hdu, **kw)
        sc.allow_huge_operations = True
        t = best(lambda: np.asarray(sc.spectral_smooth(k1).unmasked_data[:]), n=2)
        print("spectral_smooth 33 taps | 128x128x128 | %s | %.4f | %.2f" % (label, t, vox / t / 1e6))
        t = best(lambda: np.asarray(sc.spatial_smooth(k2).unmasked_data[:]), n=1)
        print("spatial_smooth 29x29 | 128x128x128 | %s | %.4f | %.2f" % (label, t, vox / t / 1e6))
    t = best(lambda: O.spectral_smooth(data, None, k1.array), n=2)
    print("spectral_smooth 33 taps | 128x128x128 | numpy oracle (1 thread) | %.4f | %.2f" % (t, vox / t / 1e6))
    t = best(lambda: O.spatial_smooth(data[:16], None, k2.array), n=1)
    print("spatial_smooth 29x29 | 16x128x128 | numpy oracle (1 thread) | %.4f | %.2f" % (t, 16 * 128 * 128 / t / 1e6))


if __name__ == "__main__":
    main()
