"""Synthetic orbit catalogue with the regime mix of the reference's sampler.

The reference ships `envs/1.5_hour_viz_20000_of_20000_sample_orbits_seed_0.npy` (20000x6,
m and m/s, GCRS), produced by envs/orbit_gen.py from `init_state_vec` (dynamics.py:357-399):
regimes LEO / MEO / GEO / Tundra / Molniya in proportion 2:2:2:1:1, uniformly random
inc/raan/argp/nu, half of the GEO rows exactly circular-equatorial.  That file is an input of
the reference repo and does not travel with this package; `synthetic_catalogue` draws rows
from the same distributions (vectorised; not the same random stream) so that benchmarks and
tests exercise the same branch mix -- including the exactly circular / equatorial rows that
take rv2coe's special branches (farnocchia.py:278-309).
"""
import numpy as np

MU = 398600441800000.0
RE_EQ = 6378136.6   # poliastro Earth.R used by the reference sampler


def coe2rv_host(p, ecc, inc, raan, argp, nu):
    cn, sn = np.cos(nu), np.sin(nu)
    fr, fv = p / (1 + ecc * cn), np.sqrt(MU / p)
    px, py, vx, vy = cn * fr, sn * fr, -sn * fv, (ecc + cn) * fv
    cO, sO, ci, si, cw, sw = np.cos(raan), np.sin(raan), np.cos(inc), np.sin(inc), np.cos(argp), np.sin(argp)
    r00, r01 = cO * cw - sO * ci * sw, -cO * sw - sO * ci * cw
    r10, r11 = sO * cw + cO * ci * sw, -sO * sw + cO * ci * cw
    r20, r21 = si * sw, si * cw
    return np.stack([px * r00 + py * r01, px * r10 + py * r11, px * r20 + py * r21,
                     vx * r00 + vy * r01, vx * r10 + vy * r11, vx * r20 + vy * r21], axis=-1)


def synthetic_catalogue(n=20000, seed=0):
    rs = np.random.RandomState(seed)
    regime = rs.choice(5, size=n, p=[0.25, 0.25, 0.25, 0.125, 0.125])   # LEO MEO GEO Tundra Molniya
    inc = np.radians(rs.uniform(0, 180, n))
    raan = np.radians(rs.uniform(0, 360, n))
    argp = np.radians(rs.uniform(0, 360, n))
    nu = np.radians(rs.uniform(0, 360, n))
    a = np.empty(n)
    ecc = np.empty(n)
    for k, (lo, hi) in enumerate([(RE_EQ + 300e3, RE_EQ + 2000e3), (RE_EQ + 2000e3, RE_EQ + 35786e3)]):
        idx = np.where(regime == k)[0]
        aa, ee = rs.uniform(lo, hi, idx.size), rs.uniform(0, .25, idx.size)
        bad = aa * np.sqrt(1 - ee ** 2) <= RE_EQ + 300e3
        while bad.any():   # exo-atmospheric rejection (dynamics.py:369-382)
            aa[bad], ee[bad] = rs.uniform(lo, hi, bad.sum()), rs.uniform(0, .25, bad.sum())
            bad = aa * np.sqrt(1 - ee ** 2) <= RE_EQ + 300e3
        a[idx], ecc[idx] = aa, ee
    g = regime == 2
    stationary = rs.randint(0, 2, g.sum())
    a[g], ecc[g], inc[g] = 42164e3, stationary * rs.uniform(0, .25, g.sum()), 0.0
    t = regime == 3
    a[t], inc[t], ecc[t], argp[t] = 42164e3, np.radians(63.4), 0.2, np.radians(270)
    mo = regime == 4
    a[mo], inc[mo], ecc[mo], argp[mo] = 26600e3, np.radians(63.4), 0.737, np.radians(270)
    return np.ascontiguousarray(coe2rv_host(a * (1 - ecc ** 2), ecc, inc, raan, argp, nu))
