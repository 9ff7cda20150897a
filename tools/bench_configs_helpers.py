import ctypes as C
from spectral_cube_amd import _lib


def replicate_planes(dev, tile):
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
