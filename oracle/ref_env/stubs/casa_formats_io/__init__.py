"""Stub of `casa_formats_io` (absent here; CASA I/O is out of scope)."""


def _no(*a, **k):
    raise NotImplementedError("casa_formats_io stub")


getdesc = coordsys_to_astropy_wcs = image_to_dask = _no
