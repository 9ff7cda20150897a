"""Import the *reference* spectral_cube inside the build container.

TEST INFRASTRUCTURE ONLY (never imported by the product path, never shipped
to the GPU box).  Run with ``/opt/conda/bin/python3.9 -B`` (``-B`` keeps
__pycache__ out of the read-only /root/reference tree).  Recipe from
SURVEY.md section 8(c):

* numpy 1.26 removed aliases astropy 4.3.1 still touches -> re-add them;
* `radio_beam`, `casa_formats_io` are absent -> tiny stubs (beam/CASA only);
* astropy 4.3.1 has no StokesCoord -> dummies (Stokes container only).

None of the shims touch moment / smoothing / interpolation arithmetic.
"""
import os
import sys
import warnings

REFERENCE = os.environ.get("SPC_REFERENCE", "/root/reference")


def load_reference():
    import numpy as np
    for name, val in dict(asscalar=lambda a: a.item(), alen=len, float=float,
                          int=int, bool=bool, object=object, complex=complex,
                          str=str, msort=lambda a: np.sort(a, axis=0)).items():
        if not hasattr(np, name):
            setattr(np, name, val)
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "stubs"))
    warnings.simplefilter("ignore")
    import astropy.coordinates as ac

    class _Any:
        def __init__(self, *a, **k):
            pass

    for nm in ("StokesCoord", "custom_stokes_symbol_mapping", "StokesSymbol"):
        if not hasattr(ac, nm):
            setattr(ac, nm, _Any)
    sys.dont_write_bytecode = True
    sys.path.insert(0, REFERENCE)
    import spectral_cube
    return spectral_cube


if __name__ == "__main__":
    sc = load_reference()
    print("reference imported from", sc.__file__)
