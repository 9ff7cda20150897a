"""Minimal FITS WCS for the hot path (pure numpy; astropy is not required on
the GPU box).

Covers what the kernels' host-side inputs need:
a LINEAR spectral axis and a celestial pair with CDELT / PCi_j / CDi_j /
CROTA2 linear part and a zenithal (TAN, SIN, ARC, STG, ZEA) or CAR
projection (FITS WCS paper II).  Validated against astropy.wcs through the
committed vectors in tests/golden/wcs.npz.

It stands in for the astropy.wcs calls made by
``spectral_cube/base_class.py:178-241`` (world), ``spectral_cube.py:1455-1535``
(_pix_cen, _pix_size_slice) and ``wcs_utils.py:28-45`` (drop_axis).
"""
import copy
import re

import numpy as np

_D2R = np.pi / 180.0
_R2D = 180.0 / np.pi
_ZENITHAL = ("TAN", "SIN", "ARC", "STG", "ZEA")
# cylindrical / pseudo-cylindrical projections with the reference point at native (0, 0) (FITS paper II sections 5.2 - 5.3)
_CYLINDRICAL = ("CAR", "SFL", "CEA", "MER", "AIT")
_PROJ_CODE = {"TAN": 0, "SIN": 1, "ARC": 2, "STG": 3, "ZEA": 4, "CAR": 5, "SFL": 6, "CEA": 7, "MER": 8, "AIT": 9}


_AXIS_KEY = re.compile(r"^(?:(?:CTYPE|CRVAL|CRPIX|CDELT|CUNIT|CROTA|NAXIS|CNAME|CRDER|CSYER)(\d)|(?:PC|CD)(\d)_(\d)|"
                       r"PC00(\d)00(\d)|(?:PV|PS)(\d)_\d+)$")
_SPECTRAL_ALGORITHMS = ("F2W", "F2V", "F2A", "W2F", "W2V", "W2A", "V2F", "V2W", "V2A", "A2F", "A2W", "A2V", "LOG", "TAB", "GRI", "GRA")
_SPECTRAL_KEYS = ("RESTFRQ", "RESTFREQ", "RESTWAV", "SPECSYS", "SSYSOBS", "VELREF", "VELOSYS", "ZSOURCE", "SSYSSRC")


def key_axes(key):
    """FITS axis numbers a WCS keyword refers to: CTYPE3 -> {3}, PC1_3 -> {1, 3}, PV2_3 -> {2} (the 3 is a parameter
    index), A_3_0 / RESTFRQ -> {} (not an axis keyword)"""
    m = _AXIS_KEY.match(str(key).upper())
    return {int(g) for g in m.groups() if g is not None} if m else set()


def parse_header(header):
    """Accept a dict-like, an astropy Header, or FITS header text (80-char
    cards or newline separated ``KEY = value / comment`` lines)."""
    if header is None:
        return {}
    if isinstance(header, SimpleWCS):
        return dict(header.header)
    if hasattr(header, "keys") and not isinstance(header, str):
        return {str(k).upper(): header[k] for k in header.keys() if k not in ("", "COMMENT", "HISTORY")}
    text = str(header)
    if "\n" in text:
        cards = text.split("\n")
    else:
        cards = [text[i:i + 80] for i in range(0, len(text), 80)]
    out = {}
    for card in cards:
        if "=" not in card[:10] and "=" not in card.split("/")[0]:
            continue
        key, _, rest = card.partition("=")
        key = key.strip().upper()
        if not key or key in ("COMMENT", "HISTORY", "END"):
            continue
        rest = rest.strip()
        if rest.startswith("'"):
            end = rest.find("'", 1)
            val = rest[1:end].strip() if end > 0 else rest[1:].strip()
        else:
            val = rest.split("/")[0].strip()
            if val in ("T", "F"):
                val = (val == "T")
            else:
                try:
                    val = int(val)
                except ValueError:
                    try:
                        val = float(val.replace("D", "E"))
                    except ValueError:
                        pass
        out[key] = val
    return out


# ---- keywords that alter the celestial transform ------------------------------------------------------
# reproject hands complete astropy.wcs.WCS objects to reproject_interp (spectral_cube.py:2700-2732), i.e. wcslib's
# core transformation PLUS astropy's distortion stages (all_pix2world / all_world2pix).  What SimpleWCS models of that
# is listed here; a keyword of these families that it does not model RAISES - a header is never silently read as a
# simpler one (checked against astropy 4.3.1 in oracle/gen_golden.py::case_wcs_strict):
#   modelled   CTYPE / CRVAL / CRPIX / CDELT / CUNIT (deg, arcmin, arcsec, mas, rad) / PCi_j / CDi_j / CROTA2,
#              LONPOLE / LATPOLE, PV1_0..PV1_4 (fiducial offset, phi_0, theta_0, LONPOLE, LATPOLE on the longitude axis),
#              PV2_1 (CEA), the SIP polynomials A_p_q / B_p_q (-SIP suffix or not: astropy applies them either way)
#   ignored like astropy ignores them   AP_p_q / BP_p_q (the inverse polynomials: all_world2pix inverts A / B itself),
#              A_DMAX / B_DMAX, CROTA1, CROTA2 beside PCi_j or CDi_j, CDi_j beside PCi_j, alternate descriptions (CTYPE1A ...)
#   refused    PV2_m on projections that take no parameter (a TAN header with PV cards is SCAMP's TPV distortion in newer
#              astropy), slant SIN, PSi_m, CPDIS / CQDIS / DPj / DQj look-up distortions, D2IM* detector corrections,
#              projections / units / axis orders outside the lists above
_SIP_COEF = re.compile(r"^(A|B|AP|BP)_(\d+)_(\d+)$")
_SIP_META = re.compile(r"^(A|B|AP|BP)_(ORDER|DMAX)$")
_PV_KEY = re.compile(r"^(PV|PS)(\d+)_(\d+)$")
_REFUSED = re.compile(r"^(CPDIS\d|CQDIS\d|CPERR\d|CQERR\d|DP\d|DQ\d|D2IMDIS\d|D2IMERR\d|D2IMEXT|D2IM\d|AXISCORR|DVERR\d)")
_ANGLE_TO_DEG = {"": 1.0, "deg": 1.0, "degree": 1.0, "degrees": 1.0, "arcmin": 1.0 / 60.0, "arcsec": 1.0 / 3600.0,
                 "mas": 1.0 / 3.6e6, "rad": 180.0 / np.pi}
SIP_MAX_ORDER = 9            # (the limit of astropy's Sip / of spc_celestial_wcs)


def _sip_index(p, q):
    """position of the coefficient of u^p v^q in the triangular tables of spc_celestial_wcs (row p holds 10 - p entries)"""
    return p * (SIP_MAX_ORDER + 1) - p * (p - 1) // 2 + q


def _poly(c, u, v):
    """sum_p sum_q c[p, q] u^p v^q (Horner in both variables; c is (n + 1, n + 1), zero where p + q > n)"""
    n = c.shape[0] - 1
    out = np.zeros(np.broadcast(u, v).shape)
    for p_ in range(n, -1, -1):
        row = np.zeros_like(out)
        for q_ in range(n - p_, -1, -1):
            row = row * v + c[p_, q_]
        out = out * u + row
    return out


def _poly_grad(c, u, v):
    """(d/du, d/dv) of _poly"""
    n = c.shape[0] - 1
    cu = np.zeros_like(c)
    cv = np.zeros_like(c)
    for p_ in range(n + 1):
        for q_ in range(n + 1 - p_):
            if p_ >= 1:
                cu[p_ - 1, q_] = p_ * c[p_, q_]
            if q_ >= 1:
                cv[p_, q_ - 1] = q_ * c[p_, q_]
    return _poly(cu, u, v), _poly(cv, u, v)


class SimpleWCS:
    """3-axis (x, y, spectral) or 2-axis (x, y) FITS WCS.

    strict=True (default): a header keyword that changes the celestial transform and is not modelled raises
    NotImplementedError here; strict=False (the WCS a cube is READ with: its moments along the spectral axis need no
    celestial transform) records it and raises at the first celestial use instead."""

    def __init__(self, header=None, naxis=None, strict=True):
        h = parse_header(header)
        self.header = h
        n = int(naxis or h.get("WCSAXES", h.get("NAXIS", 3)))
        self.naxis = n = min(n, 3)
        g = h.get
        self.ctype = [str(g("CTYPE%d" % (i + 1), "")) for i in range(n)]
        self.cunit = [str(g("CUNIT%d" % (i + 1), "")).strip() for i in range(n)]
        self.crval = np.array([float(g("CRVAL%d" % (i + 1), 0.0)) for i in range(n)])
        self.crpix = np.array([float(g("CRPIX%d" % (i + 1), 0.0)) for i in range(n)])
        self.cdelt = np.array([float(g("CDELT%d" % (i + 1), 1.0)) for i in range(n)])
        pc = np.eye(n)
        has_pc = any(("PC%d_%d" % (i + 1, j + 1)) in h or ("PC%03d%03d" % (i + 1, j + 1)) in h for i in range(n) for j in range(n))
        has_cd = any(("CD%d_%d" % (i + 1, j + 1)) in h for i in range(n) for j in range(n))
        if has_cd and not has_pc:            # (wcslib: PCi_j wins when a header carries both)
            cd = np.zeros((n, n))
            for i in range(n):
                for j in range(n):
                    cd[i, j] = float(g("CD%d_%d" % (i + 1, j + 1), 0.0))
            self.cdelt = np.ones(n)
            pc = cd
        else:
            for i in range(n):
                for j in range(n):
                    pc[i, j] = float(g("PC%d_%d" % (i + 1, j + 1), g("PC%03d%03d" % (i + 1, j + 1), pc[i, j])))
            if "CROTA2" in h and not any(k.startswith("PC") for k in h):
                rho = float(h["CROTA2"]) * _D2R
                lam = self.cdelt[1] / self.cdelt[0]
                pc[0, 0], pc[0, 1] = np.cos(rho), -lam * np.sin(rho)
                pc[1, 0], pc[1, 1] = np.sin(rho) / lam, np.cos(rho)
        self.pc = pc
        self.shape_hint = tuple(int(g("NAXIS%d" % (i + 1), 0)) for i in range(n))[::-1]
        self.proj = self.ctype[0][5:8] if len(self.ctype[0]) >= 8 else ""
        if self.proj and self.proj not in _PROJ_CODE:
            raise NotImplementedError("projection %r not supported by SimpleWCS (built: %s)" % (self.proj, ", ".join(sorted(_PROJ_CODE))))
        self._unsupported = []
        celestial = n >= 2 and bool(self.proj)
        # CEA: PV2_1 = lambda (cos^2 of the standard parallel), default 1 (Lambert)
        self.pv1 = float(g("PV2_1", 1.0)) if self.proj == "CEA" else 1.0
        if self.proj == "SIN" and (float(g("PV2_1", 0.0)) != 0.0 or float(g("PV2_2", 0.0)) != 0.0):
            raise NotImplementedError("slant orthographic projection (SIN with PV2_1 / PV2_2) is not built")
        # PV1_m on the longitude axis (FITS paper II section 2.5; wcslib wcsset / celset): m = 0 fiducial offset flag,
        # 1, 2 = native coordinates (phi_0, theta_0) of the fiducial point, 3, 4 = LONPOLE, LATPOLE (these win over the
        # keywords of that name)
        self.lonpole = g("PV1_3", g("LONPOLE", None))
        self.latpole = float(g("PV1_4", g("LATPOLE", 90.0)))
        native0 = (0.0, 90.0) if self.proj in _ZENITHAL else (0.0, 0.0)
        self.user_fiducial = ("PV1_1" in h) or ("PV1_2" in h)
        self.phi0 = float(g("PV1_1", native0[0]))
        self.theta0 = float(g("PV1_2", native0[1]))
        self.offset = bool(float(g("PV1_0", 0.0)) != 0.0) and self.user_fiducial
        self.x0 = self.y0 = 0.0
        self.sip_a = self.sip_b = None
        self.frame = _celestial_frame(self.ctype[:2], h) if celestial else None
        if celestial:
            self._celestial_units()
            self._scan_keywords()
        if self._unsupported and strict:
            raise NotImplementedError(self._unsupported[0])
        self._setup_pole()
        if celestial and self.offset:
            # the projection plane is shifted so that the fiducial point (phi_0, theta_0) lands on (0, 0) (wcslib prjoff)
            x0, y0 = self._native_to_plane(np.float64(self.phi0 * _D2R), np.float64(self.theta0 * _D2R))
            self.x0, self.y0 = float(x0), float(y0)

    # -- what the header asks for beyond the linear + spherical part -----------------------------------
    def _celestial_units(self):
        """celestial CUNITs other than degrees are scaled to degrees (what wcsset does with them)"""
        for i in (0, 1):
            u_ = self.cunit[i]
            f = _ANGLE_TO_DEG.get(u_, _ANGLE_TO_DEG.get(u_.lower()))
            if f is None:
                self._unsupported.append("celestial CUNIT%d = %r: deg, arcmin, arcsec, mas or rad" % (i + 1, u_))
                continue
            if f != 1.0:
                self.crval = self.crval.copy()
                self.crval[i] *= f
                if np.all(self.cdelt == 1.0) and any(("CD%d_%d" % (i + 1, j + 1)) in self.header for j in range(self.naxis)) \
                        and not any(k.startswith("PC") for k in self.header):
                    self.pc = self.pc.copy()
                    self.pc[i, :] *= f
                else:
                    self.cdelt = self.cdelt.copy()
                    self.cdelt[i] *= f
                self.cunit = list(self.cunit)
                self.cunit[i] = "deg"

    def _scan_keywords(self):
        h, bad = self.header, self._unsupported
        for i in (0, 1):
            c = self.ctype[i]
            tail = c[8:] if len(c) > 8 else ""
            if tail not in ("", "-SIP"):
                bad.append("CTYPE%d = %r: distortion code %r is not built (only -SIP)" % (i + 1, c, tail))
        if self.ctype[1][5:8] != self.proj or len(self.ctype[1]) < 8:
            bad.append("CTYPE1 = %r / CTYPE2 = %r: the two celestial axes name different projections" % (self.ctype[0], self.ctype[1]))
        lead = (self.ctype[0][:4].upper(), self.ctype[1][:4].upper())
        if lead[0] in ("DEC-",) or lead[0].endswith("LAT") or lead[1] in ("RA--",) or lead[1].endswith("LON"):
            bad.append("CTYPE1 = %r / CTYPE2 = %r: latitude before longitude is not built (axis 1 must be the longitude)"
                       % (self.ctype[0], self.ctype[1]))
        orders, coefs = {}, {}
        for key, val in h.items():
            k = str(key).upper()
            if _REFUSED.match(k):
                bad.append("%s: look-up / detector distortions (astropy's cpdis / det2im stages) are not built" % k)
                continue
            m = _PV_KEY.match(k)
            if m:
                kind, ax, idx = m.group(1), int(m.group(2)), int(m.group(3))
                if ax > 2:
                    continue                       # (the spectral axis: see spectral_pix2world)
                if kind == "PS":
                    bad.append("%s: string-valued projection parameters are not built" % k)
                elif ax == 1 and idx > 4:
                    bad.append("%s: longitude-axis parameters beyond PV1_4 are not built (TPV / SCAMP distortion?)" % k)
                elif ax == 2 and not (self.proj == "CEA" and idx == 1) and float(val) != 0.0:
                    bad.append("%s = %r: projection %s takes no such parameter here (a -TAN header with PV cards is SCAMP's TPV "
                               "distortion in newer astropy; ZPN / AZP / SZP / COP ... are not built)" % (k, val, self.proj))
                continue
            m = _SIP_META.match(k)
            if m:
                if m.group(2) == "ORDER":
                    orders[m.group(1)] = int(val)
                continue
            m = _SIP_COEF.match(k)
            if m:
                coefs.setdefault(m.group(1), {})[(int(m.group(2)), int(m.group(3)))] = float(val)
        # SIP (Shupe et al. 2005) exactly as astropy reads it (astropy/wcs/wcs.py::_read_sip_kw): the forward polynomials
        # need A_ORDER > 1 and B_ORDER > 1; terms with p + q > order are not read
        na, nb = orders.get("A"), orders.get("B")
        if (na is None) != (nb is None):
            bad.append("A_ORDER / B_ORDER: SIP needs both")
        elif na is None:
            if coefs.get("A") or coefs.get("B"):
                bad.append("SIP coefficients A_p_q / B_p_q without A_ORDER / B_ORDER")
        elif na > 1 or nb > 1:
            if not (na > 1 and nb > 1):
                bad.append("A_ORDER = %d, B_ORDER = %d: astropy reads SIP only when both exceed 1" % (na, nb))
            elif max(na, nb) > SIP_MAX_ORDER:
                bad.append("SIP order %d > %d" % (max(na, nb), SIP_MAX_ORDER))
            else:
                n_ = max(na, nb)
                a, b = np.zeros((n_ + 1, n_ + 1)), np.zeros((n_ + 1, n_ + 1))
                for (tab, lim, name) in ((a, na, "A"), (b, nb, "B")):
                    for (p_, q_), v in coefs.get(name, {}).items():
                        if p_ + q_ <= lim:
                            tab[p_, q_] = v
                if np.any(a) or np.any(b):
                    self.sip_a, self.sip_b = a, b

    def _require_celestial(self):
        if self.naxis < 2 or not self.proj:
            raise ValueError("WCS does not contain two spatial axes.")
        if self._unsupported:
            raise NotImplementedError(self._unsupported[0])

    # -- FITS paper II section 2.4: celestial coordinates of the native pole
    def _setup_pole(self):
        if self.naxis < 2 or not self.proj:
            return
        a0, d0 = self.crval[0] * _D2R, self.crval[1] * _D2R
        th0 = self.theta0 * _D2R
        ph0 = self.phi0 * _D2R
        if self.lonpole is None:
            lonpole = self.phi0 + (0.0 if d0 >= th0 else 180.0)
        else:
            lonpole = float(self.lonpole)
        php = lonpole * _D2R
        if abs(th0 - np.pi / 2) < 1e-12:
            ap, dp = a0, d0
        else:
            ct0, st0 = np.cos(th0), np.sin(th0)
            dphi = php - ph0
            base = np.arctan2(st0, ct0 * np.cos(dphi))
            den = np.sqrt(1.0 - (ct0 * np.sin(dphi)) ** 2)
            arg = np.clip(np.sin(d0) / den, -1.0, 1.0)
            cands = [base + np.arccos(arg), base - np.arccos(arg)]
            cands = [(c + np.pi) % (2 * np.pi) - np.pi for c in cands]
            cands = [c for c in cands if -np.pi / 2 - 1e-12 <= c <= np.pi / 2 + 1e-12]
            if not cands:
                raise ValueError("invalid LONPOLE/LATPOLE for this header")
            dp = min(cands, key=lambda c: abs(c - self.latpole * _D2R))
            if abs(abs(dp) - np.pi / 2) < 1e-12:
                ap = a0 + (dphi - np.pi if dp > 0 else -dphi)
            else:
                sa = np.sin(dphi) * ct0 / np.cos(d0)
                ca = (st0 - np.sin(dp) * np.sin(d0)) / (np.cos(dp) * np.cos(d0))
                ap = a0 - np.arctan2(sa, ca)
        self._ap, self._dp, self._php = ap, dp, php

    # ------------------------------------------------------------------
    @property
    def pixel_scale_matrix(self):
        return self.cdelt[:, None] * self.pc

    def _lin2(self):
        return (self.cdelt[:, None] * self.pc)[:2, :2]

    def sip_tables(self):
        """(order, A table, B table): the SIP polynomials in the triangular layout of spc_celestial_wcs, order 0 = none"""
        ta, tb = np.zeros(_sip_index(SIP_MAX_ORDER, 0) + 1), np.zeros(_sip_index(SIP_MAX_ORDER, 0) + 1)
        if self.sip_a is None:
            return 0, ta, tb
        n_ = self.sip_a.shape[0] - 1
        for p_ in range(n_ + 1):
            for q_ in range(n_ + 1 - p_):
                ta[_sip_index(p_, q_)] = self.sip_a[p_, q_]
                tb[_sip_index(p_, q_)] = self.sip_b[p_, q_]
        return n_, ta, tb

    def celestial_params(self):
        """the numbers spc_wcs_pixel_map_f64 needs (include/spcube_hip.h: spc_celestial_wcs):
        (proj code, crpix (x, y), lin 2x2, lin^-1 2x2, alpha_p, delta_p, phi_p, pv1, (x0, y0), (sip order, A, B))"""
        self._require_celestial()
        code = _PROJ_CODE[self.proj]
        m = self._lin2()
        return (code, (float(self.crpix[0]), float(self.crpix[1])), tuple(m.ravel()), tuple(np.linalg.inv(m).ravel()),
                float(self._ap), float(self._dp), float(self._php), float(self.pv1), (float(self.x0), float(self.y0)),
                self.sip_tables())

    # -- projection plane <-> native sphere (FITS paper II section 5), radians in / degrees on the plane ---------
    def _plane_to_native(self, x, y):
        if self.proj in _CYLINDRICAL:
            with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
                if self.proj == "CAR":
                    phi, theta = x * _D2R, y * _D2R
                elif self.proj == "SFL":
                    theta = y * _D2R
                    c = np.cos(theta)
                    phi = np.where(c != 0.0, x * _D2R / np.where(c != 0.0, c, 1.0), np.where(x == 0.0, 0.0, np.nan))
                elif self.proj == "CEA":
                    s_ = self.pv1 * y * _D2R
                    theta = np.where(np.abs(s_) <= 1.0 + 1e-13, np.arcsin(np.clip(s_, -1.0, 1.0)), np.nan)
                    phi = x * _D2R
                elif self.proj == "MER":
                    theta = 2.0 * np.arctan(np.exp(y * _D2R)) - np.pi / 2
                    phi = x * _D2R
                else:  # AIT (Hammer-Aitoff)
                    xr, yr = x * _D2R, y * _D2R
                    z2 = 1.0 - (xr / 4.0) ** 2 - (yr / 2.0) ** 2
                    zz = np.sqrt(np.where(z2 >= 0.5 - 1e-13, np.maximum(z2, 0.5), np.nan))     # outside the ellipse: not on the sky
                    phi = 2.0 * np.arctan2(zz * xr / 2.0, 2.0 * zz * zz - 1.0)
                    theta = np.arcsin(np.clip(yr * zz, -1.0, 1.0))
                # the native sphere ends at |phi| = 180, |theta| = 90 (wcslib's bounds check: such pixels are not on the sky)
                bad = (np.abs(phi) > np.pi * (1 + 1e-12)) | (np.abs(theta) > np.pi / 2 * (1 + 1e-12))
                phi = np.where(bad, np.nan, phi)
        else:
            r = np.hypot(x, y)
            phi = np.arctan2(x, -y)
            rr = r * _D2R
            if self.proj == "TAN":
                theta = np.arctan2(1.0, rr)
            elif self.proj == "SIN":
                theta = np.arccos(np.clip(rr, -1.0, 1.0))
            elif self.proj == "ARC":
                theta = np.pi / 2 - rr
            elif self.proj == "STG":
                theta = np.pi / 2 - 2.0 * np.arctan(rr / 2.0)
            else:  # ZEA
                theta = np.pi / 2 - 2.0 * np.arcsin(np.clip(rr / 2.0, -1.0, 1.0))
        return phi, theta

    def _native_vec_to_plane(self, phi, rho, zn):
        """native longitude phi and the (cos theta, sin theta) = (rho, zn) pair of a unit vector -> plane (degrees)"""
        with np.errstate(invalid="ignore", divide="ignore"):
            if self.proj in _CYLINDRICAL:
                phi = np.mod(phi + np.pi, 2 * np.pi) - np.pi
                theta = np.arctan2(zn, rho)
                if self.proj == "CAR":
                    x, y = phi * _R2D, theta * _R2D
                elif self.proj == "SFL":
                    x, y = phi * np.cos(theta) * _R2D, theta * _R2D
                elif self.proj == "CEA":
                    x, y = phi * _R2D, _R2D * np.sin(theta) / self.pv1
                elif self.proj == "MER":
                    x, y = phi * _R2D, np.where(np.abs(theta) < np.pi / 2, _R2D * np.log(np.tan(np.pi / 4 + theta / 2)), np.nan)
                else:  # AIT
                    gam = _R2D * np.sqrt(2.0 / (1.0 + np.cos(theta) * np.cos(phi / 2.0)))
                    x, y = 2.0 * gam * np.cos(theta) * np.sin(phi / 2.0), gam * np.sin(theta)
            else:
                if self.proj == "TAN":
                    r = np.where(zn > 0, _R2D * rho / zn, np.nan)
                elif self.proj == "SIN":
                    r = np.where(zn >= 0, _R2D * rho, np.nan)
                elif self.proj == "ARC":
                    r = _R2D * np.arctan2(rho, zn)
                elif self.proj == "STG":
                    r = 2.0 * _R2D * rho / (1.0 + zn)
                else:  # ZEA
                    r = 2.0 * _R2D * rho / np.sqrt(2.0 * (1.0 + zn))
                x, y = r * np.sin(phi), -r * np.cos(phi)
        return x, y

    def _native_to_plane(self, phi, theta):
        return self._native_vec_to_plane(phi, np.cos(theta), np.sin(theta))

    def celestial_pix2world(self, px, py):
        """0-based pixel -> (lon, lat) degrees (astropy's all_pix2world: SIP, then wcslib's core)."""
        self._require_celestial()
        px = np.asarray(px, dtype=np.float64)
        py = np.asarray(py, dtype=np.float64)
        m = self._lin2()
        dx, dy = px + 1.0 - self.crpix[0], py + 1.0 - self.crpix[1]
        if self.sip_a is not None:
            dx, dy = dx + _poly(self.sip_a, dx, dy), dy + _poly(self.sip_b, dx, dy)
        x = m[0, 0] * dx + m[0, 1] * dy + self.x0
        y = m[1, 0] * dx + m[1, 1] * dy + self.y0
        phi, theta = self._plane_to_native(x, y)
        ap, dp, php = self._ap, self._dp, self._php
        dphi = phi - php
        st, ct = np.sin(theta), np.cos(theta)
        # celestial unit vector about the native pole's meridian; the latitude from atan2 of its components (asin of the
        # third alone loses half the digits near the celestial poles - wcslib's sphx2s switches to acos there)
        xc = st * np.cos(dp) - ct * np.sin(dp) * np.cos(dphi)
        yc = -ct * np.sin(dphi)
        zc = st * np.sin(dp) + ct * np.cos(dp) * np.cos(dphi)
        lon = ap + np.arctan2(yc, xc)
        lat = np.arctan2(zc, np.hypot(xc, yc))
        lon = np.mod(lon * _R2D, 360.0)
        return lon, lat * _R2D

    def _sip_invert(self, uu, vv, maxiter=50, tol=1e-13):
        """(u, v) with u + A(u, v) = uu, v + B(u, v) = vv: Newton from (uu, vv).  astropy's all_world2pix inverts the
        forward polynomials too (fixed-point, stopped at 1e-4 pixel by default); this is that iteration's limit.
        Points where it does not converge (far outside the image, where the polynomial folds over) come back NaN."""
        u, v = np.array(uu, dtype=np.float64, copy=True), np.array(vv, dtype=np.float64, copy=True)
        with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
            done = np.zeros(u.shape, dtype=bool)
            for _ in range(maxiter):
                f = u + _poly(self.sip_a, u, v) - uu
                g_ = v + _poly(self.sip_b, u, v) - vv
                au, av = _poly_grad(self.sip_a, u, v)
                bu, bv = _poly_grad(self.sip_b, u, v)
                j00, j01, j10, j11 = 1.0 + au, av, bu, 1.0 + bv
                det = j00 * j11 - j01 * j10
                du = (j11 * f - j01 * g_) / det
                dv = (j00 * g_ - j10 * f) / det
                u = np.where(done, u, u - du)
                v = np.where(done, v, v - dv)
                done |= (np.abs(du) <= tol * np.maximum(1.0, np.abs(u))) & (np.abs(dv) <= tol * np.maximum(1.0, np.abs(v)))
                if done.all():
                    break
            u = np.where(done, u, np.nan)
            v = np.where(done, v, np.nan)
        return u, v

    def celestial_world2pix(self, lon, lat):
        """(lon, lat) degrees -> 0-based pixel; NaN where not projectable (astropy's all_world2pix)."""
        self._require_celestial()
        lon = np.asarray(lon, dtype=np.float64) * _D2R
        lat = np.asarray(lat, dtype=np.float64) * _D2R
        ap, dp, php = self._ap, self._dp, self._php
        da = lon - ap
        sl, cl = np.sin(lat), np.cos(lat)
        # native unit vector (accurate near the native pole, unlike asin())
        xn = -cl * np.sin(da)
        yn = sl * np.cos(dp) - cl * np.sin(dp) * np.cos(da)
        zn = sl * np.sin(dp) + cl * np.cos(dp) * np.cos(da)
        rho = np.hypot(xn, yn)
        phi = php + np.arctan2(xn, yn)
        x, y = self._native_vec_to_plane(phi, rho, zn)
        x, y = x - self.x0, y - self.y0
        minv = np.linalg.inv(self._lin2())
        dx = minv[0, 0] * x + minv[0, 1] * y
        dy = minv[1, 0] * x + minv[1, 1] * y
        if self.sip_a is not None:
            dx, dy = self._sip_invert(dx, dy)
        return dx + self.crpix[0] - 1.0, dy + self.crpix[1] - 1.0

    # -- spectral ---------------------------------------------------------
    def _require_linear_spectral(self):
        """the spectral axis is read as LINEAR in its CTYPE3 quantity.  wcslib (which the reference's world / reproject
        calls go through) also knows non-linear axes - CTYPE3 = 'VOPT-F2W' (optical velocity sampled evenly in frequency),
        '-LOG', '-TAB', the grism codes, and AIPS' 'FELO-xxx' which it translates to VOPT-F2W: those are refused, not read
        as linear."""
        c = str(self.ctype[2]).strip().upper()
        if c.startswith("FELO") or (len(c) >= 8 and c[4] == "-" and c[5:8] in _SPECTRAL_ALGORITHMS):
            raise NotImplementedError("CTYPE3 = %r: non-linear spectral axes (wcslib's spectral algorithm codes) are not built"
                                      % self.ctype[2])

    def spectral_pix2world(self, pz):
        if self.naxis < 3:
            raise ValueError("no spectral axis")
        self._require_linear_spectral()
        pz = np.asarray(pz, dtype=np.float64)
        if abs(self.pc[2, 0]) + abs(self.pc[2, 1]) + abs(self.pc[0, 2]) + abs(self.pc[1, 2]) > 0:
            raise NotImplementedError("spectral and celestial axes must be separable")
        return self.crval[2] + self.cdelt[2] * self.pc[2, 2] * (pz + 1.0 - self.crpix[2])

    def spectral_world2pix(self, w):
        """world coordinate (in this axis' CUNIT3) -> 0-based channel (fractional); linear axis"""
        if self.naxis < 3:
            raise ValueError("no spectral axis")
        self._require_linear_spectral()
        w = np.asarray(w, dtype=np.float64)
        return (w - self.crval[2]) / (self.cdelt[2] * self.pc[2, 2]) + self.crpix[2] - 1.0

    @property
    def spectral_unit(self):
        return self.cunit[2] if self.naxis >= 3 else ""

    def drop_spectral(self):
        """``wcs_utils.drop_axis(wcs, 2)`` (wcs_utils.py:28-45)."""
        out = copy.deepcopy(self)
        out.naxis = 2
        out.ctype, out.cunit = out.ctype[:2], out.cunit[:2]
        out.crval, out.crpix, out.cdelt = out.crval[:2], out.crpix[:2], out.cdelt[:2]
        out.pc = out.pc[:2, :2]
        out.header = {k: v for k, v in self.header.items() if 3 not in key_axes(k)}
        return out

    def with_spectral(self, crval, cdelt, crpix=1.0, cunit=None):
        out = copy.deepcopy(self)
        out.crval = out.crval.copy(); out.cdelt = out.cdelt.copy(); out.crpix = out.crpix.copy()
        out.crval[2], out.cdelt[2], out.crpix[2] = crval, cdelt, crpix
        out.pc = out.pc.copy(); out.pc[2, 2] = 1.0
        if cunit is not None:
            out.cunit = list(out.cunit); out.cunit[2] = cunit
        out.header = {k: v for k, v in out.header.items() if k not in ("PC3_3", "CD3_3", "PC003003")}   # folded into CDELT3
        out.header.update(CRVAL3=float(crval), CDELT3=float(cdelt), CRPIX3=float(crpix))
        if cunit is not None:
            out.header["CUNIT3"] = cunit
        return out

    def to_header(self):
        h = {"WCSAXES": self.naxis}
        for i in range(self.naxis):
            h["CTYPE%d" % (i + 1)] = self.ctype[i]
            h["CUNIT%d" % (i + 1)] = self.cunit[i]
            h["CRVAL%d" % (i + 1)] = float(self.crval[i])
            h["CRPIX%d" % (i + 1)] = float(self.crpix[i])
            h["CDELT%d" % (i + 1)] = float(self.cdelt[i])
            for j in range(self.naxis):
                if self.pc[i, j] != (1.0 if i == j else 0.0):
                    h["PC%d_%d" % (i + 1, j + 1)] = float(self.pc[i, j])
        return h


# ---- celestial reference frames ---------------------------------------------------------------------
# reproject_interp hands full WCS objects to astropy (spectral_cube.py:2700-2732), which turns the target's
# pixels into sky coordinates IN THE TARGET'S FRAME, transforms them to the source's frame and only then asks
# the source WCS for pixels.  The frame of a header follows astropy.wcs.utils.wcs_to_celestial_frame; the
# transformations between ICRS, FK5(equinox), FK4-NO-E(equinox) and Galactic are constant rotations of the unit sphere
# (restated from their published definitions below); FK4 adds the E-terms of aberration on its side.
def _celestial_frame(ctype, header):
    """('icrs',) | ('fk5', equinox_jyear) | ('fk4', equinox_byear) | ('galactic',) | ('other', lon, lat)"""
    x, y = str(ctype[0])[:4].upper(), str(ctype[1])[:4].upper()
    g = header.get
    if x == "RA--" and y == "DEC-":
        radesys = str(g("RADESYS", g("RADECSYS", "")) or "").strip().upper()
        eq = g("EQUINOX", g("EPOCH", None))                       # wcslib reads EPOCH as the old name of EQUINOX
        eq = None if eq is None or eq == "" else float(eq)
        if radesys == "":
            radesys = "ICRS" if eq is None else ("FK4" if eq < 1984.0 else "FK5")
        if radesys == "ICRS":
            return ("icrs",)
        if radesys == "FK5":
            return ("fk5", 2000.0 if eq is None else eq)
        if radesys in ("FK4", "FK4-NO-E"):
            return (radesys.lower(), 1950.0 if eq is None else eq)
        return ("other", "RA--:" + radesys, y)
    if x == "GLON" and y == "GLAT":
        return ("galactic",)
    return ("other", x, y)


def _rot(angle_deg, axis):
    """passive rotation about a coordinate axis (the convention of astropy's rotation_matrix)"""
    a = np.deg2rad(angle_deg)
    c, s_ = np.cos(a), np.sin(a)
    i = "xyz".index(axis)
    a1, a2 = (i + 1) % 3, (i + 2) % 3
    r = np.zeros((3, 3))
    r[i, i] = 1.0
    r[a1, a1], r[a1, a2], r[a2, a1], r[a2, a2] = c, s_, -s_, c
    return r


def _precess_from_j2000(jyear):
    """precession J2000 -> Julian epoch *jyear*: Capitaine et al. 2003 as printed in USNO Circular 179 (the IAU
    2006 angles zeta, z, theta in arcseconds) - what astropy's FK5 frame uses"""
    t = (jyear - 2000.0) / 100.0
    zeta = np.polyval((-0.0000003173, -0.000005971, 0.01801828, 0.2988499, 2306.083227, 2.650545), t) / 3600.0
    z = np.polyval((-0.0000002904, -0.000028596, 0.01826837, 1.0927348, 2306.077181, -2.650545), t) / 3600.0
    theta = np.polyval((-0.0000001274, -0.000007089, -0.04182264, -0.4294934, 2004.191903, 0.0), t) / 3600.0
    return _rot(-z, "z") @ _rot(theta, "y") @ _rot(-zeta, "z")


# ICRS -> FK5(J2000): the frame bias (eta0, xi0, dalpha0 in milliarcseconds; Hilton & Hohenkerk 2004)
_ICRS_TO_FK5J2000 = _rot(19.9 / 3600000.0, "x") @ _rot(9.1 / 3600000.0, "y") @ _rot(-22.9 / 3600000.0, "z")
# FK5(J2000) -> Galactic: north galactic pole and the longitude of the north celestial pole in FK5 J2000
# (the IAU 1958 definition carried from B1950 to J2000)
_FK5J2000_TO_GAL = (_rot(180.0 - 122.9319185680026, "z") @ _rot(90.0 - 27.12825118085622, "y")
                    @ _rot(192.8594812065348, "z"))


# ---- FK4 (Besselian equinoxes) ------------------------------------------------------------------------
# What astropy.coordinates does for RADESYS = 'FK4' / 'FK4-NO-E' (or RA/DEC with EQUINOX < 1984), restated from the
# published definitions it cites: FK4 carries the elliptic ("E") terms of aberration, a small vector removed from /
# added to the unit vector (Explanatory Supplement 1992 / Seidelmann; max 0.34 arcsec); without them the frame is
# related to FK5 J2000 by Standish's (1982) matrix plus Murray's (1989, eq. 29) rotation-rate correction at the
# epoch of observation (= the equinox for a header, which carries no other), to Galactic by the IAU 1958 pole in
# B1950, and precessed with Newcomb's angles.
def _byear_to_jd(byear):
    return 2415020.31352 + (float(byear) - 1900.0) * 365.242198781         # (ERFA epb2jd)


def _precess_newcomb(b1, b2):
    """precession between two Besselian equinoxes, Newcomb (Explanatory Supplement 1961 / Kinoshita): angles in arcsec"""
    t1 = (b1 - 1850.0) / 1000.0
    dt = (b2 - 1850.0) / 1000.0 - t1
    zeta = np.polyval((17.995, 30.240 - 0.27 * t1, 23035.545 + t1 * 139.720 + 0.060 * t1 * t1, 0.0), dt) / 3600.0
    z = np.polyval((18.325, 109.480 + 0.39 * t1, 23035.545 + t1 * 139.720 + 0.060 * t1 * t1, 0.0), dt) / 3600.0
    theta = np.polyval((-41.8, -42.65 - 0.37 * t1, 20051.12 - 85.29 * t1 - 0.37 * t1 * t1, 0.0), dt) / 3600.0
    return _rot(-z, "z") @ _rot(theta, "y") @ _rot(-zeta, "z")


# FK4 (no E-terms) B1950 -> FK5 J2000 (Standish 1982) and its drift per Julian century of the observation epoch (Murray 1989)
_B1950_TO_J2000 = np.array([[0.9999256794956877, -0.0111814832204662, -0.0048590038153592],
                            [0.0111814832391717, 0.9999374848933135, -0.0000271625947142],
                            [0.0048590037723143, -0.0000271702937440, 0.9999881946023742]])
_FK4_DRIFT = np.array([[-0.0026455262, -1.1539918689, +2.1111346190],
                       [+1.1540628161, -0.0129042997, +0.0236021478],
                       [-2.1112979048, -0.0056024448, +0.0102587734]]) * 1.0e-6
# FK4 (no E-terms) B1950 -> Galactic: IAU 1958 (pole 12h49m, +27.4 deg; longitude of the celestial pole 123 deg)
_FK4B1950_TO_GAL = _rot(180.0 - 123.0, "z") @ _rot(90.0 - 27.4, "y") @ _rot(192.25, "z")


def _fk4_to_fk5j2000(byear):
    jyear_obs = 2000.0 + (_byear_to_jd(byear) - 2451545.0) / 365.25       # epoch of observation = the equinox
    return (_B1950_TO_J2000 + _FK4_DRIFT * ((jyear_obs - 1950.0) / 100.0)) @ _precess_newcomb(float(byear), 1950.0)


def fk4_e_terms(byear):
    """the E-terms of aberration vector at a Besselian equinox (constant of aberration 20.49552 arcsec = 0.0056932 deg;
    eccentricity and longitude of perigee of the solar orbit, obliquity IAU 1980: Explanatory Supplement 1992)"""
    jd = _byear_to_jd(byear)
    t50 = (jd - _byear_to_jd(1950.0)) / 36525.0
    k = np.radians(0.0056932)
    e = np.polyval((-0.000000126, -0.00004193, 0.01673011), t50)
    g = np.radians(np.polyval((0.012, 1.65, 6190.67, 1015489.951), t50) / 3600.0)
    t2k = (jd - 2451545.0) / 36525.0
    o = np.radians(np.polyval((0.001813, -0.00059, -46.8150, 84381.448), t2k) / 3600.0)
    return np.array([e * k * np.sin(g), -e * k * np.cos(g) * np.cos(o), -e * k * np.cos(g) * np.sin(o)])


def _linear(frame):
    """the frame without its E-terms (FK4 -> FK4-NO-E at the same equinox)"""
    return ("fk4-no-e", float(frame[1])) if frame[0] == "fk4" else tuple(frame)


def _to_fk5j2000(frame):
    if frame[0] == "icrs":
        return _ICRS_TO_FK5J2000
    if frame[0] == "galactic":
        return _FK5J2000_TO_GAL.T
    if frame[0] == "fk5":
        return _precess_from_j2000(2000.0) @ _precess_from_j2000(float(frame[1])).T
    if frame[0] == "fk4-no-e":
        return _fk4_to_fk5j2000(float(frame[1]))
    raise NotImplementedError("celestial frame %r: ICRS, FK5, FK4, FK4-NO-E and Galactic are related to each other here "
                              "(ecliptic / helioprojective / unnamed headers reproject only onto their own frame - astropy's "
                              "wcs_to_celestial_frame knows no others either)" % (frame,))


def _linear_rotation(a, b):
    """3 x 3 matrix between two frames without E-terms, along the route astropy's transformation graph takes"""
    if a == b:
        return np.eye(3)
    if a[0] == "fk4-no-e" and b[0] == "fk4-no-e":
        return _precess_newcomb(a[1], b[1])
    if a[0] == "fk4-no-e" and b[0] == "galactic":              # the IAU 1958 definition itself, not via FK5
        return _FK4B1950_TO_GAL @ _precess_newcomb(a[1], 1950.0)
    if a[0] == "galactic" and b[0] == "fk4-no-e":
        return (_FK4B1950_TO_GAL @ _precess_newcomb(b[1], 1950.0)).T
    if b[0] == "fk4-no-e":
        # FK5 J2000 -> FK4-NO-E(equinox): Newcomb's precession FROM B1950 (his polynomials are not their own inverse: the
        # transpose of the matrix TO B1950 differs by microarcseconds), after the transposed Standish / Murray matrix
        jyear_obs = 2000.0 + (_byear_to_jd(b[1]) - 2451545.0) / 365.25
        back = _precess_newcomb(1950.0, float(b[1])) @ (_B1950_TO_J2000 + _FK4_DRIFT * ((jyear_obs - 1950.0) / 100.0)).T
        return back @ _to_fk5j2000(a)
    return _to_fk5j2000(b).T @ _to_fk5j2000(a)


def frame_transform(frame_from, frame_to):
    """how unit vectors of *frame_from* become unit vectors of *frame_to*: None when the two are the same frame, else
    (remove, rot, add): subtract the E-terms vector `remove` (v - D + (D.v) v, renormalised; None = no such step), rotate by
    the 3 x 3 `rot`, add the E-terms `add` (ten fixed-point rounds of v = (D + v0) / (1 + D.v), renormalised; None = no such
    step) - FK4 on either side brings its step, everything else is a rotation."""
    if frame_from is None or frame_to is None or tuple(frame_from) == tuple(frame_to):
        return None
    a, b = _linear(frame_from), _linear(frame_to)
    rot = _linear_rotation(a, b)                # NotImplementedError for frames that are not built
    remove = fk4_e_terms(frame_from[1]) if frame_from[0] == "fk4" else None
    add = fk4_e_terms(frame_to[1]) if frame_to[0] == "fk4" else None
    return remove, rot, add


def frame_rotation(frame_from, frame_to):
    """3 x 3 matrix taking unit vectors of *frame_from* to *frame_to* (frames without E-terms), None for the same frame"""
    tr = frame_transform(frame_from, frame_to)
    if tr is None:
        return None
    if tr[0] is not None or tr[2] is not None:
        raise ValueError("FK4 frames are not a pure rotation: use frame_transform")
    return tr[1]


def apply_frame_transform(tr, v):
    """unit vectors v (3, ...) through frame_transform's (remove, rot, add)"""
    remove, rot, add = tr
    if remove is not None:
        d = remove.reshape((3,) + (1,) * (v.ndim - 1))
        v = v - d + np.sum(d * v, axis=0) * v
        v = v / np.sqrt(np.sum(v * v, axis=0))
    v = np.tensordot(rot, v, axes=1)
    if add is not None:
        d = add.reshape((3,) + (1,) * (v.ndim - 1))
        v0 = v
        for _ in range(10):
            v = (d + v0) / (1.0 + np.sum(d * v, axis=0))
        v = v / np.sqrt(np.sum(v * v, axis=0))
    return v


_SPECTRAL_SI = {"m/s": ("speed", 1.0), "km/s": ("speed", 1e3), "cm/s": ("speed", 1e-2),
                "Hz": ("freq", 1.0), "kHz": ("freq", 1e3), "MHz": ("freq", 1e6), "GHz": ("freq", 1e9),
                "m": ("length", 1.0), "cm": ("length", 1e-2), "mm": ("length", 1e-3), "um": ("length", 1e-6),
                "nm": ("length", 1e-9), "Angstrom": ("length", 1e-10), "angstrom": ("length", 1e-10)}


def spectral_unit_scale(unit_from, unit_to):
    """factor that takes a spectral coordinate in *unit_from* to *unit_to* (what wcslib's SI normalisation
    does for two headers of one spectral type, spectral_cube.py:218-228); ValueError when the two are not
    the same kind of quantity - that is a change of spectral representation, not a regrid."""
    a, b = (unit_from or "").strip(), (unit_to or "").strip()
    if a == b:
        return 1.0
    if a not in _SPECTRAL_SI or b not in _SPECTRAL_SI or _SPECTRAL_SI[a][0] != _SPECTRAL_SI[b][0]:
        raise ValueError("cannot relate spectral units %r and %r: convert the cube's spectral axis first" % (a, b))
    return _SPECTRAL_SI[a][1] / _SPECTRAL_SI[b][1]


def join_celestial_spectral(celestial, spectral, nz=None):
    """3-axis WCS with the celestial axes of *celestial* and the spectral axis of *spectral* (keywords are chosen by
    the axis they name, not by the digits they contain: PV2_3 and the SIP terms A_3_0 stay celestial, the PC1_3 /
    PC3_1 cross terms of either header are dropped - the two parts are separable by construction)"""
    h = {k: v for k, v in celestial.header.items() if 3 not in key_axes(k) and k not in _SPECTRAL_KEYS}
    for k, v in spectral.header.items():
        if key_axes(k) == {3} and not k.startswith("NAXIS"):
            h[k] = v
        elif k in _SPECTRAL_KEYS:
            h[k] = v
    h["NAXIS"] = 3
    h["WCSAXES"] = 3
    if nz is None:
        nz = spectral.header.get("NAXIS3")
    if nz is not None:
        h["NAXIS3"] = int(nz)
    return SimpleWCS(h)


def check_same_spectral_kind(src, dst):
    """reproject's spectral regrid is LINEAR in the axis both headers describe: it is only a regrid when they describe
    the same axis.  The reference hands both full WCSs to reproject_interp, where wcslib converts between spectral
    representations (spectral_cube.py:2700-2732); that conversion is not built, so a different CTYPE3 (VRAD vs VOPT,
    both in m/s), rest frequency / wavelength or SPECSYS raises instead of resampling at the wrong channels."""
    hs, hd = src.header, dst.header
    a, b = str(hs.get("CTYPE3", "")).strip().upper()[:4], str(hd.get("CTYPE3", "")).strip().upper()[:4]
    if a and b and a != b:
        raise NotImplementedError("reproject onto another spectral representation (CTYPE3 %r -> %r) is not built: convert "
                                  "the cube's spectral axis first" % (hs.get("CTYPE3"), hd.get("CTYPE3")))
    for keys in (("RESTFRQ", "RESTFREQ"), ("RESTWAV",)):
        va = next((float(hs[k]) for k in keys if k in hs and hs[k] not in ("", None)), None)
        vb = next((float(hd[k]) for k in keys if k in hd and hd[k] not in ("", None)), None)
        if va is not None and vb is not None and va != 0.0 and vb != 0.0 and abs(va - vb) > 1e-9 * abs(va):
            raise NotImplementedError("reproject between rest %s %r and %r is not built" % (keys[0], va, vb))
    sa, sb = str(hs.get("SPECSYS", "")).strip().upper(), str(hd.get("SPECSYS", "")).strip().upper()
    if sa and sb and sa != sb:
        raise NotImplementedError("reproject between spectral reference frames (SPECSYS %s -> %s) is not built" % (sa, sb))


def angular_separation(lon1, lat1, lon2, lat2):
    """Vincenty formula, radians (astropy.coordinates.angular_separation, used
    by spectral_cube.py:1482-1486)."""
    sdlon, cdlon = np.sin(lon2 - lon1), np.cos(lon2 - lon1)
    slat1, slat2, clat1, clat2 = np.sin(lat1), np.sin(lat2), np.cos(lat1), np.cos(lat2)
    num1 = clat2 * sdlon
    num2 = clat1 * slat2 - slat1 * clat2 * cdlon
    den = slat1 * slat2 + clat1 * clat2 * cdlon
    return np.arctan2(np.hypot(num1, num2), den)


def pix_cen_spatial(wcs, shape):
    """y and x offset maps of ``_pix_cen`` (spectral_cube.py:1476-1496):
    cumulative angular separations (degrees) from pixel 0 along each axis."""
    nz, ny, nx = shape
    yy, xx = np.mgrid[0:ny, 0:nx]
    lon, lat = wcs.celestial_pix2world(xx, yy)
    lon, lat = lon * _D2R, lat * _D2R
    dx = angular_separation(lon[:, :-1], lat[:, :-1], lon[:, 1:], lat[:, :-1])
    dy = angular_separation(lon[:-1, :], lat[:-1, :], lon[1:, :], lat[1:, :])
    x = np.zeros((ny, nx))
    y = np.zeros((ny, nx))
    x[:, 1:] = np.cumsum(np.degrees(dx), axis=1)
    y[1:, :] = np.cumsum(np.degrees(dy), axis=0)
    return y, x


def pix_size(wcs, axis):
    """``_pix_size_slice`` (spectral_cube.py:1510-1535) in the header's units."""
    psm = wcs.pixel_scale_matrix
    if axis == 0:
        return abs(psm[2, 2])
    if axis in (1, 2):
        return float(np.sum(psm[2 - axis, :] ** 2) ** 0.5)
    raise ValueError("Cubes have 3 axes.")


def reproject_pixel_map(wcs_in, wcs_out, shape_out):
    """source pixel coordinates (xs, ys) in *wcs_in* of every output pixel of
    *wcs_out*: what reproject_interp computes through astropy.wcs
    (pixel_to_world on the target, world_to_pixel on the source)."""
    ny, nx = shape_out
    yy, xx = np.mgrid[0:ny, 0:nx]
    lon, lat = wcs_out.celestial_pix2world(xx, yy)
    tr = frame_transform(wcs_out.frame, wcs_in.frame)         # NotImplementedError for pairs that are not built
    if tr is not None:
        lo, la = lon * _D2R, lat * _D2R
        v = np.stack([np.cos(la) * np.cos(lo), np.cos(la) * np.sin(lo), np.sin(la)])
        w = apply_frame_transform(tr, v)
        lon = np.mod(np.arctan2(w[1], w[0]) * _R2D, 360.0)
        lat = np.arctan2(w[2], np.hypot(w[0], w[1])) * _R2D
    return wcs_in.celestial_world2pix(lon, lat)
