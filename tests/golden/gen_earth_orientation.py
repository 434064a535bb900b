"""Generator of the Earth-orientation data and goldens (build container only; /opt/conda/bin/python3.9 has pyerfa):

    /opt/conda/bin/python3.9 tests/golden/gen_earth_orientation.py

Writes
  ssa-gym_amd/data/xys06a_halfday.npz   X, Y, s of erfa.xys06a (pyerfa, third party) on a half-day TT grid
  ssa-gym_amd/data/eop_c04.npz          x, y, UT1-UTC, dX, dY columns of the IERS 14 C04 file vendored by the reference
                                        (hpiers.obspm.fr/iers/eop/eopc04/eopc04_IAU2000.62-now; IERS data, not source)
  tests/golden/earth_orientation_golden.npz
        epochs + matrices of the REFERENCE's own gcrs2irts_matrix_b (envs/transformations.py:143-214) at random
        epochs over the whole EOP span, erfa.dat at every month start 1962-2020, and spot values of
        erfa.era00 / sp00 / pom00 / c2ixys / xys06a for the restated closed-form pieces.
"""
import os
import sys
from datetime import datetime, timedelta

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import _ref_loader as L  # noqa: E402

import erfa  # noqa: E402

tr, dy = L.load_transformations_and_dynamics()
eop = L.load_eops()
DATA = os.path.join(ROOT, "ssa-gym_amd", "data")

# ---- product tables
mjd_first, mjd_last = int(eop.index[0]), int(eop.index[-1])
assert np.all(np.diff(eop.index.values) == 1)
np.savez_compressed(os.path.join(DATA, "eop_c04.npz"), mjd0=np.int64(mjd_first), x=eop["x"].values, y=eop["y"].values,
                    UT1_UTC=eop["UT1-UTC"].values, dX=eop["dX"].values, dY=eop["dY"].values)
step = 0.5
mjd0 = mjd_first - 8.0
n = int(round((mjd_last + 16.0 - mjd0) / step))
grid = mjd0 + np.arange(n) * step
x, y, s = erfa.xys06a(2400000.5, grid)
np.savez(os.path.join(DATA, "xys06a_halfday.npz"), mjd0=np.float64(mjd0), step=np.float64(step), xys=np.stack([x, y, s], 1))
print("xys table", n, "rows", grid[0], grid[-1], "eop", mjd_first, mjd_last)

# ---- goldens
rs = np.random.RandomState(20261004)
t_lo, t_hi = datetime(1962, 1, 2), datetime(2020, 6, 22)
span = int((t_hi - t_lo).total_seconds())
secs = np.sort(rs.randint(0, span, size=400))
epochs = [t_lo + timedelta(seconds=int(k)) for k in secs]
# leap-second boundaries and day boundaries
for d in (datetime(1972, 6, 30, 23, 59, 59), datetime(1972, 7, 1, 0, 0, 0), datetime(2016, 12, 31, 23, 59, 59),
          datetime(2017, 1, 1, 0, 0, 0), datetime(1965, 12, 31, 12, 0, 0), datetime(1968, 2, 1, 0, 0, 1),
          datetime(2000, 1, 1, 12, 0, 0), datetime(2007, 4, 5, 12, 0, 0), datetime(2020, 5, 4, 0, 0, 0)):
    epochs.append(d)
M = np.array(tr.gcrs2irts_matrix_b(epochs, eop)).reshape(len(epochs), 3, 3)
ep_arr = np.array([[e.year, e.month, e.day, e.hour, e.minute, e.second] for e in epochs], dtype=np.int64)
months = [(yr, mo) for yr in range(1962, 2021) for mo in range(1, 13)]
dat_tab = np.array([[yr, mo, erfa.dat(yr, mo, 1, 0.0), erfa.dat(yr, mo, 15, 0.75)] for yr, mo in months])
tt = 37700.0 + rs.uniform(size=64) * (59000 - 37700)
xs, ys, ss = erfa.xys06a(2400000.5, tt)
ut = rs.uniform(-1e-5, 1.0, size=64)
dj = 2400000.5 + np.floor(tt)
era = erfa.era00(dj, ut)
sp = erfa.sp00(2400000.5, tt)
xp, yp = rs.normal(size=64) * 1e-6, rs.normal(size=64) * 1e-6
pom = np.array([erfa.pom00(a, b, c) for a, b, c in zip(xp, yp, sp)])
c2i = np.array([erfa.c2ixys(a, b, c) for a, b, c in zip(xs, ys, ss)])
np.savez_compressed(os.path.join(HERE, "earth_orientation_golden.npz"), epochs=ep_arr, matrices=M, dat=dat_tab,
                    tt=tt, xys=np.stack([xs, ys, ss], 1), ut=ut, dj=dj, era=era, sp=sp, xp=xp, yp=yp, pom=pom, c2i=c2i)
print("golden: %d epochs" % len(epochs))
