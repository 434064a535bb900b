"""Native GCRS->ITRS matrix builder: the init-time work of the reference's `gcrs2irts_matrix_b`
(envs/transformations.py:143-214) and `get_eops` (:19-31) without liberfa, pandas or a network.

The reference evaluates, per epoch, ERFA `cal2jd, dat, xys06a, c2ixys, era00, cr, rz, pom00, sp00, rxr` and
interpolates the IERS C04 Earth-orientation parameters linearly in the day fraction.  Here:

  * calendar / leap seconds / Earth rotation angle / TIO locator / polar motion / CIP->matrix are restated
    from the published SOFA algorithms (closed-form, a few lines each) in numpy, vectorised over epochs;
  * the CIP coordinates X, Y and the CIO locator s of IAU 2006/2000A (`xys06a`: a 1 365 + 687-term nutation
    series plus polynomial precession -- the one piece that is not a few lines) come from a table of
    xys06a values on a HALF-DAY TT grid covering 1962-01-01 .. 2020-07-08 (`data/xys06a_halfday.npy`,
    generated once by tests/golden/gen_earth_orientation.py with pyerfa) through 10-point Lagrange
    interpolation.  Interpolation error, measured against direct xys06a evaluations at random epochs:
    < 7e-15 rad in X, Y and < 2e-19 rad in s (the shortest nutation periods, 5-7 days, are sampled 10-14
    times per period) -- five orders below the 1e-11 agreement asserted for the final matrices;
  * the EOP series x, y, UT1-UTC, dX, dY is the IERS 14 C04 solution 1962-01-01 .. 2020-06-23
    (`data/eop_c04.npz`, columns of the file the reference vendors), or any IERS C04 text file given by path /
    $SSA_GYM_EOP, parsed by `read_eop_c04`.

Pinned by tests/test_earth_orientation.py against matrices the reference's own gcrs2irts_matrix_b produced at
random epochs of the whole EOP span, against the shipped 2020-05-04 tables and against the SOFA cookbook
matrix the reference holds (tests.py:107-109).
"""
import os
from datetime import datetime

import numpy as np

DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")
DAYSEC = 86400.0
DAS2R = 4.848136811095359935899141e-6     # arc seconds -> radians (ERFA_DAS2R)
D2PI = 6.283185307179586476925287
DJ00, DJC, DJM0 = 2451545.0, 36525.0, 2400000.5
TT_MINUS_TAI = 32.184

# ---- eraDat: TAI-UTC.  1972 onward: whole leap seconds (IERS Bulletin C); 1960-1971: the rubber-second era,
# delta = base + (MJD - ref) * rate.  (year, month, delta, ref MJD, rate s/day)
_LEAP = [
    (1960, 1, 1.4178180, 37300.0, 0.0012960), (1961, 1, 1.4228180, 37300.0, 0.0012960),
    (1961, 8, 1.3728180, 37300.0, 0.0012960), (1962, 1, 1.8458580, 37665.0, 0.0011232),
    (1963, 11, 1.9458580, 37665.0, 0.0011232), (1964, 1, 3.2401300, 38761.0, 0.0012960),
    (1964, 4, 3.3401300, 38761.0, 0.0012960), (1964, 9, 3.4401300, 38761.0, 0.0012960),
    (1965, 1, 3.5401300, 38761.0, 0.0012960), (1965, 3, 3.6401300, 38761.0, 0.0012960),
    (1965, 7, 3.7401300, 38761.0, 0.0012960), (1965, 9, 3.8401300, 38761.0, 0.0012960),
    (1966, 1, 4.3131700, 39126.0, 0.0025920), (1968, 2, 4.2131700, 39126.0, 0.0025920),
    (1972, 1, 10.0, 0, 0), (1972, 7, 11.0, 0, 0), (1973, 1, 12.0, 0, 0), (1974, 1, 13.0, 0, 0), (1975, 1, 14.0, 0, 0),
    (1976, 1, 15.0, 0, 0), (1977, 1, 16.0, 0, 0), (1978, 1, 17.0, 0, 0), (1979, 1, 18.0, 0, 0), (1980, 1, 19.0, 0, 0),
    (1981, 7, 20.0, 0, 0), (1982, 7, 21.0, 0, 0), (1983, 7, 22.0, 0, 0), (1985, 7, 23.0, 0, 0), (1988, 1, 24.0, 0, 0),
    (1990, 1, 25.0, 0, 0), (1991, 1, 26.0, 0, 0), (1992, 7, 27.0, 0, 0), (1993, 7, 28.0, 0, 0), (1994, 7, 29.0, 0, 0),
    (1996, 1, 30.0, 0, 0), (1997, 7, 31.0, 0, 0), (1999, 1, 32.0, 0, 0), (2006, 1, 33.0, 0, 0), (2009, 1, 34.0, 0, 0),
    (2012, 7, 35.0, 0, 0), (2015, 7, 36.0, 0, 0), (2017, 1, 37.0, 0, 0),
]


def cal2jd(iy, im, id_):
    """eraCal2jd: Gregorian calendar date -> (2400000.5, MJD at 0h); integer arithmetic truncates toward zero as in C."""
    def q(a, b):
        return int(a / b)
    my = q(im - 14, 12)
    iypmy = iy + my
    djm = float(q(1461 * (iypmy + 4800), 4) + q(367 * (im - 2 - 12 * my), 12) - q(3 * q(iypmy + 4900, 100), 4) + id_ - 2432076)
    return DJM0, djm


def dat(iy, im, id_, fd):
    """eraDat: TAI-UTC [s] for a UTC date and day fraction."""
    if iy < 1960:
        raise ValueError("UTC is not defined before 1960")
    _, djm = cal2jd(iy, im, id_)
    m = 12 * iy + im
    row = None
    for r in _LEAP:
        if m >= 12 * r[0] + r[1]:
            row = r
    da = row[2]
    if row[4]:
        da += (djm + fd - row[3]) * row[4]
    return da


def _rz(psi, r):
    """eraRz: r <- Rz(psi) r (the reference's eraRZ, transformations.py:97-114), batched over leading axes."""
    s, c = np.sin(psi)[..., None], np.cos(psi)[..., None]
    r0 = c * r[..., 0, :] + s * r[..., 1, :]
    r1 = -s * r[..., 0, :] + c * r[..., 1, :]
    return np.stack([r0, r1, r[..., 2, :]], axis=-2)


def _ry(theta, r):
    s, c = np.sin(theta)[..., None], np.cos(theta)[..., None]
    r0 = c * r[..., 0, :] - s * r[..., 2, :]
    r2 = s * r[..., 0, :] + c * r[..., 2, :]
    return np.stack([r0, r[..., 1, :], r2], axis=-2)


def _rx(phi, r):
    s, c = np.sin(phi)[..., None], np.cos(phi)[..., None]
    r1 = c * r[..., 1, :] + s * r[..., 2, :]
    r2 = -s * r[..., 1, :] + c * r[..., 2, :]
    return np.stack([r[..., 0, :], r1, r2], axis=-2)


def _eye(n):
    return np.broadcast_to(np.eye(3), (n, 3, 3)).copy()


def era00(dj1, dj2):
    """eraEra00: Earth rotation angle (IAU 2000) of the UT1 date dj1 + dj2, radians in [0, 2 pi)."""
    dj1, dj2 = np.asarray(dj1, dtype=np.float64), np.asarray(dj2, dtype=np.float64)
    d1, d2 = np.minimum(dj1, dj2), np.maximum(dj1, dj2)
    t = d1 + (d2 - DJ00)
    f = np.fmod(d1, 1.0) + np.fmod(d2, 1.0)
    w = np.fmod(D2PI * (f + 0.7790572732640 + 0.00273781191135448 * t), D2PI)
    return np.where(w < 0, w + D2PI, w)


def sp00(date1, date2):
    """eraSp00: TIO locator s' (IERS 2003), radians."""
    t = ((np.asarray(date1, dtype=np.float64) - DJ00) + np.asarray(date2, dtype=np.float64)) / DJC
    return -47e-6 * t * DAS2R


def pom00(xp, yp, sp):
    """eraPom00: polar-motion matrix Rx(-yp) Ry(-xp) Rz(sp)."""
    xp = np.atleast_1d(np.asarray(xp, dtype=np.float64))
    r = _rz(np.atleast_1d(sp), _eye(len(xp)))
    r = _ry(-xp, r)
    return _rx(-np.atleast_1d(yp), r)


def c2ixys(x, y, s):
    """eraC2ixys: celestial-to-intermediate matrix from the CIP X, Y and the CIO locator s."""
    x, y, s = (np.atleast_1d(np.asarray(v, dtype=np.float64)) for v in (x, y, s))
    r2 = x * x + y * y
    e = np.where(r2 > 0.0, np.arctan2(y, x), 0.0)
    d = np.arctan(np.sqrt(r2 / (1.0 - r2)))
    r = _rz(e, _eye(len(x)))
    r = _ry(d, r)
    return _rz(-(e + s), r)


# ---- X, Y, s table (half-day TT grid) and its interpolation
_XYS = None
XYS_ORDER = 10


def _xys_table():
    global _XYS
    if _XYS is None:
        z = np.load(os.path.join(DATA, "xys06a_halfday.npz"))
        _XYS = (float(z["mjd0"]), float(z["step"]), np.ascontiguousarray(z["xys"]))
    return _XYS


def xys06a_interp(tt_mjd):
    """(X, Y, s) of IAU 2006/2000A at TT = 2400000.5 + tt_mjd by 10-point Lagrange interpolation of the table."""
    mjd0, step, tab = _xys_table()
    t = np.atleast_1d(np.asarray(tt_mjd, dtype=np.float64))
    u = (t - mjd0) / step
    i0 = np.floor(u).astype(np.int64) - (XYS_ORDER // 2 - 1)
    if i0.min() < 0 or i0.max() + XYS_ORDER > tab.shape[0]:
        raise ValueError("epoch outside the X, Y, s table (TT MJD %.1f .. %.1f)" % (mjd0 + 4 * step, mjd0 + (tab.shape[0] - 5) * step))
    k = np.arange(XYS_ORDER)
    d = (u - i0)[:, None] - k[None, :]                       # distance to each node, in steps
    w = np.ones_like(d)
    for a in range(XYS_ORDER):                               # w_a = prod_{b != a} (u - x_b) / (x_a - x_b)
        for b in range(XYS_ORDER):
            if a != b:
                w[:, a] *= d[:, b] / float(a - b)
    vals = tab[i0[:, None] + k[None, :]]                     # (n, order, 3)
    out = np.einsum('na,nac->nc', w, vals)
    return out[:, 0], out[:, 1], out[:, 2]


# ---- EOP series
_EOP_COLS = ('x', 'y', 'UT1-UTC', 'dX', 'dY')


class EopTable:
    """daily IERS C04 series x, y [arcsec], UT1-UTC [s], dX, dY [arcsec] from MJD `mjd0` on (contiguous days);
    `eop[col][mjd]` indexing as on the reference's pandas frame (transformations.py:178-206)."""

    def __init__(self, mjd0, cols):
        self.mjd0 = int(mjd0)
        self.cols = {k: np.asarray(v, dtype=np.float64) for k, v in cols.items()}
        self.n = len(self.cols['x'])

    class _Col:
        def __init__(self, t, a):
            self.t, self.a = t, a

        def __getitem__(self, mjd):
            i = np.asarray(mjd).astype(np.int64) - self.t.mjd0
            if np.any(i < 0) or np.any(i >= self.t.n):
                raise KeyError("MJD %s outside the EOP series (%d .. %d)" % (mjd, self.t.mjd0, self.t.mjd0 + self.t.n - 1))
            return self.a[i]

    def __getitem__(self, col):
        return EopTable._Col(self, self.cols[col])


def read_eop_c04(path):
    """parse an IERS 'EOP (IERS) 14 C04' text file (the format the reference reads with np.genfromtxt(skip_header=14),
    transformations.py:24-31): data rows are `year month day MJD x y UT1-UTC LOD dX dY ...`."""
    rows = []
    with open(path) as f:
        for line in f:
            p = line.split()
            if len(p) >= 10 and p[0].isdigit() and len(p[0]) == 4 and p[3].isdigit():
                rows.append([float(v) for v in p[:10]])
    a = np.array(rows)
    if a.size == 0 or np.any(np.diff(a[:, 3]) != 1):
        raise ValueError("%s: no contiguous daily C04 rows found" % path)
    return EopTable(int(a[0, 3]), {'x': a[:, 4], 'y': a[:, 5], 'UT1-UTC': a[:, 6], 'dX': a[:, 8], 'dY': a[:, 9]})


_EOP = None


def builtin_eop():
    """the shipped IERS 14 C04 series (1962-01-01 .. 2020-06-23)."""
    global _EOP
    if _EOP is None:
        z = np.load(os.path.join(DATA, "eop_c04.npz"))
        _EOP = EopTable(int(z["mjd0"]), {k: z[k.replace('-', '_')] for k in _EOP_COLS})
    return _EOP


def gcrs2irts_matrix_b(t, eop=None):
    """GCRS->ITRS matrices (ITRS = M @ GCRS) for a datetime or a sequence of datetimes (UTC), computed as the reference's
    gcrs2irts_matrix_b does (transformations.py:143-214), vectorised.  Returns (3,3) or (n,3,3)."""
    single = isinstance(t, datetime)
    ts = [t] if single else list(t)
    eop = builtin_eop() if eop is None else eop
    n = len(ts)
    date = np.empty(n)
    frac = np.empty(n)
    leap = np.empty(n)
    for k, ti in enumerate(ts):                              # calendar + leap seconds per epoch (integer work)
        _, date[k] = cal2jd(ti.year, ti.month, ti.day)
        frac[k] = (60.0 * (60.0 * ti.hour + ti.minute) + ti.second) / DAYSEC        # (:169)
        leap[k] = dat(ti.year, ti.month, ti.day, frac[k])
    utc = date + frac
    tt = (utc + leap / DAYSEC) + TT_MINUS_TAI / DAYSEC                               # (:171-173)

    def lerp(col):                                                                  # (:176, :184-185, :203-204)
        return eop[col][date] * (1 - frac) + eop[col][date + 1] * frac
    tut = frac + lerp('UT1-UTC') / DAYSEC
    x, y, s = xys06a_interp(tt)
    x = x + lerp('dX') * DAS2R
    y = y + lerp('dY') * DAS2R
    rc2i = c2ixys(x, y, s)
    rc2ti = _rz(era00(DJM0 + date, tut), rc2i)
    rpom = pom00(lerp('x') * DAS2R, lerp('y') * DAS2R, sp00(DJM0, tt))
    out = np.einsum('nij,njk->nik', rpom, rc2ti)
    return out[0] if single else out
