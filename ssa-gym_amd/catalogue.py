"""Synthetic orbit catalogue by the reference's own recipe.

The reference ships `envs/1.5_hour_viz_20000_of_20000_sample_orbits_seed_0.npy` (20000x6, m and m/s, GCRS), produced by
envs/orbit_gen.py:30-70: per row a regime is drawn -- LEO / MEO / GEO / Tundra / Molniya with probabilities 1/3, 1/3, 1/9, 1/9, 1/9
(:53) -- and `init_state_vec` (dynamics.py:357-399) candidates of that regime are drawn until one is ACCEPTED (:55-70): propagated over
4 h in 150 s steps it must stay above 300 km altitude and, seen from the observer (38.83 N, 77.31 W) above 15 deg elevation, either be
visible all the time or be visible within the first 45 min and never out of sight for 1.5 h or longer.  Regime shares are therefore
exactly the draw probabilities (6 755 LEO rows in the reference file), every GEO row is equatorial (inc = 0: 2 231 rows) and half of
them circular (1 135: `stationary * ecc`, :383-385), eccentricities reach 0.737 (Molniya); WITHIN a regime the rule shapes the
distribution of inclination, node and phase (a LEO object has to pass over the site on consecutive revolutions).  That file is an
input of the reference repo and does not travel with this package; `synthetic_catalogue` draws rows by the same recipe (vectorised:
not the same random stream), including the exactly circular / equatorial rows that take rv2coe's special branches
(farnocchia.py:278-309).  Workload synthesis for bench.py and the tests -- numpy, no device needed: Kepler's equation in closed
elliptic form for the 96 sample times, the elevation of envs/transformations.py:330-352.
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


def _draw_elements(rs, regime, k):
    """k candidates of one regime, as init_state_vec (dynamics.py:357-399): (a, ecc, inc, raan, argp, nu)"""
    inc = np.radians(rs.uniform(0, 180, k))
    raan = np.radians(rs.uniform(0, 360, k))
    argp = np.radians(rs.uniform(0, 360, k))
    nu = np.radians(rs.uniform(0, 360, k))
    if regime in (0, 1):      # LEO / MEO: exo-atmospheric rejection on the semi-minor axis (:369-382)
        lo, hi = [(RE_EQ + 300e3, RE_EQ + 2000e3), (RE_EQ + 2000e3, RE_EQ + 35786e3)][regime]
        a, ecc = rs.uniform(lo, hi, k), rs.uniform(0, .25, k)
        bad = a * np.sqrt(1 - ecc ** 2) <= RE_EQ + 300e3
        while bad.any():
            a[bad], ecc[bad] = rs.uniform(lo, hi, bad.sum()), rs.uniform(0, .25, bad.sum())
            bad = a * np.sqrt(1 - ecc ** 2) <= RE_EQ + 300e3
    elif regime == 2:         # GEO: inc = 0 always, ecc = 0 for half of them (:383-386)
        stationary = rs.randint(0, 2, k)
        a, ecc, inc = np.full(k, 42164e3), stationary * rs.uniform(0, .25, k), np.zeros(k)
    elif regime == 3:         # Tundra
        a, inc, ecc, argp = np.full(k, 42164e3), np.full(k, np.radians(63.4)), np.full(k, 0.2), np.full(k, np.radians(270))
    else:                     # Molniya
        a, inc, ecc, argp = np.full(k, 26600e3), np.full(k, np.radians(63.4)), np.full(k, 0.737), np.full(k, np.radians(270))
    return a, ecc, inc, raan, argp, nu


def _accepted(a, ecc, inc, raan, argp, nu, M_t, times, enu, obs_itrs, el_min, first, max_gap):
    """orbit_gen.py:55-70 for a batch of candidates: bool[k]"""
    k = len(a)
    cO, sO, ci, si, cw, sw = np.cos(raan), np.sin(raan), np.cos(inc), np.sin(inc), np.cos(argp), np.sin(argp)
    P = np.stack([cO * cw - sO * ci * sw, sO * cw + cO * ci * sw, si * sw], axis=1)          # perifocal unit vectors in GCRS
    Q = np.stack([-cO * sw - sO * ci * cw, -sO * sw + cO * ci * cw, si * cw], axis=1)
    E0 = 2.0 * np.arctan2(np.sqrt(1 - ecc) * np.sin(nu / 2), np.sqrt(1 + ecc) * np.cos(nu / 2))
    M0 = E0 - ecc * np.sin(E0)
    n = np.sqrt(MU / a ** 3)
    b = a * np.sqrt(1 - ecc ** 2)
    ok_alt = np.ones(k, dtype=bool)
    vis = np.empty((len(times), k), dtype=bool)
    for i, t in enumerate(times):
        M = M0 + n * t
        E = M + ecc * np.sin(M)
        for _ in range(12):       # Newton on Kepler's equation (ecc <= 0.737: converged to 1e-14 long before)
            E = E - (E - ecc * np.sin(E) - M) / (1 - ecc * np.cos(E))
        r = (a * (np.cos(E) - ecc))[:, None] * P + (b * np.sin(E))[:, None] * Q
        x = r @ M_t[i].T                                                                        # GCRS -> ITRS
        rn = np.linalg.norm(x, axis=1)
        lat = np.arcsin(x[:, 2] / rn)
        ok_alt &= rn - WGS84_A * (1 - WGS84_F * np.sin(lat) ** 2) > 300e3                      # (geodetic height to ~1 km)
        d = x - obs_itrs
        up = d @ enu[:, 2]
        vis[i] = np.arcsin(up / np.linalg.norm(d, axis=1)) >= el_min
    run = np.zeros(k, dtype=np.int64)
    worst = np.zeros(k, dtype=np.int64)
    for i in range(len(times)):
        run = np.where(vis[i], 0, run + 1)
        worst = np.maximum(worst, run)
    always = vis.all(axis=0)
    return ok_alt & (always | (vis[:first].any(axis=0) & (worst < max_gap)))


WGS84_A, WGS84_F = 6378137.0, 0.0033528106647474805
_CACHE = {}


def synthetic_catalogue(n=20000, seed=0, visibility=True):
    """n rows by orbit_gen.py's recipe (module docstring).  visibility=False: the regime mix and element distributions alone (every
    candidate accepted) -- the catalogue of rounds 1-3 up to its regime probabilities."""
    key = (int(n), int(seed), bool(visibility))
    if key in _CACHE:
        return _CACHE[key].copy()
    if visibility:      # the benchmark's catalogue ships as data (ten minutes of numpy to draw: LEO candidates pass the rule once in ~100)
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "synthetic_catalogue_n%d_seed%d.npy" % key[:2])
        if os.path.exists(path):
            _CACHE[key] = np.load(path)
            return _CACHE[key].copy()
    from datetime import datetime
    from .envs.transformations import trans_matrix_table
    from . import host
    rs = np.random.RandomState(seed)
    regime = rs.choice(5, size=n, p=[1 / 3, 1 / 3, 1 / 9, 1 / 9, 1 / 9])   # LEO MEO GEO Tundra Molniya (orbit_gen.py:53)
    step, duration = 150.0, 4 * 3600.0
    T = int(np.ceil(duration / step))
    times = step * np.arange(T)
    M_t = trans_matrix_table(datetime(2020, 5, 4, 0, 0, 0), step, T)
    obs_lla = np.array((38.828198, -77.305352, 20.0)) * [host.deg2rad, host.deg2rad, 1]
    enu, obs_itrs = host.enu_matrix(obs_lla), host.lla2ecef(obs_lla)
    el = np.empty((n, 6))
    for k in range(5):
        idx = np.where(regime == k)[0]
        got = 0
        while got < idx.size:
            batch = max(4096, 4 * (idx.size - got))
            cand = _draw_elements(rs, k, batch)
            ok = (_accepted(*cand, M_t, times, enu, obs_itrs, np.radians(15.0), int(45 * 60 / step), int(1.5 * 3600 / step))
                  if visibility else np.ones(batch, dtype=bool))
            sel = np.where(ok)[0][:idx.size - got]
            el[idx[got:got + sel.size]] = np.stack([c[sel] for c in cand], axis=1)
            got += sel.size
    a, ecc, inc, raan, argp, nu = el.T
    out = np.ascontiguousarray(coe2rv_host(a * (1 - ecc ** 2), ecc, inc, raan, argp, nu))
    _CACHE[key] = out
    return out.copy()


def regime_order(x, tile=4, n_xcd=8, one_tile_limit=20480):
    """a permutation of the objects (rows of x: GCRS states) that puts objects of the same regime into the same wavefronts: ascending
    semi-major axis, the sorted list dealt in chunks of one tile (4 objects = one wavefront) round-robin over the 8 XCDs' runs of tiles, so
    that every XCD gets the same share of every regime.  Why it matters (round 4, profiles/r04_sorted_tiles_experiment.txt): with the
    behaviour-faithful propagator the filters that leave the strong-elliptic regime late in an episode are the LEO objects (6 795 of 6 806;
    0 of the 13 194 others); in catalogue order they are spread over 76 % of the wavefronts, each of which then runs the conic chain
    and the jitter ladder for one or two of its four objects.  Returns order[m]: new position -> original row (m a multiple of tile * n_xcd;
    otherwise -- and beyond 20 480 objects, where a wavefront walks several tiles -- the plain sort)."""
    x = np.asarray(x)
    a = 1.0 / (2.0 / np.linalg.norm(x[:, :3], axis=1) - np.sum(x[:, 3:] ** 2, axis=1) / MU)
    L = np.argsort(a, kind="stable")
    m = len(L)
    nt = m // tile
    if m % (tile * n_xcd) or m > one_tile_limit:
        # (more than 20 480 objects: a wavefront walks several tiles, stride = the number of wavefronts -- in plain sorted order every wavefront's
        # walk then runs from the slowest regime to the fastest, the same mix for all of them)
        return L
    q = nt // n_xcd
    order = np.empty(m, dtype=np.int64)
    for c in range(nt):
        t = (c % n_xcd) * q + c // n_xcd
        order[tile * t:tile * t + tile] = L[tile * c:tile * c + tile]
    return order


def regime_order_env(x, e, n_env, tile=4):
    """the layout of env e of a vector env's batch (HotPathEngine.set_layout with several envs): the plain sort by semi-major axis, ROTATED
    by e / n_env of the env's tiles.  The batch is one launch in which a wavefront walks one tile of every env at the same position (8 envs
    x 20 000 objects: wavefront w takes tile w of each); sorted alike, a wavefront would meet the slow regime in all of its tiles or in none
    -- rotated, every walk runs through the same mix (the single env's rule beyond 20 480 objects, regime_order).  Returns order[m]."""
    x = np.asarray(x)
    a = 1.0 / (2.0 / np.linalg.norm(x[:, :3], axis=1) - np.sum(x[:, 3:] ** 2, axis=1) / MU)
    L = np.argsort(a, kind="stable")
    nt = len(L) // tile
    if nt < n_env or len(L) % tile:
        return L
    return np.roll(L, -tile * ((e * nt) // n_env))
