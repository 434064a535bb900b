"""The benchmark workload's catalogue (ssa-gym_amd/catalogue.py: drawn by the reference's own recipe, envs/orbit_gen.py:30-70 + envs/dynamics.py:357-399)
against the figures of the reference's catalogue file (20 000 rows, 6 755 LEO, 2 231 equatorial, 1 135 circular, ecc <= 0.737).  CPU only."""
import numpy as np

from ssa_gym_amd import catalogue as cat_mod, host
from ssa_gym_amd.catalogue import MU, RE_EQ, synthetic_catalogue


def elements(c):
    r = np.linalg.norm(c[:, :3], axis=1)
    v2 = (c[:, 3:] ** 2).sum(1)
    a = 1.0 / (2.0 / r - v2 / MU)
    h = np.cross(c[:, :3], c[:, 3:])
    rv = (c[:, :3] * c[:, 3:]).sum(1)
    evec = (v2 - MU / r)[:, None] * c[:, :3] / MU - rv[:, None] * c[:, 3:] / MU          # (the eccentricity VECTOR, as rv2coe: exact zeros stay 1e-16)
    ecc = np.linalg.norm(evec, axis=1)
    inc = np.arccos(np.clip(h[:, 2] / np.linalg.norm(h, axis=1), -1, 1))
    return a, ecc, inc


def test_committed_catalogue_has_the_reference_files_regime_mix():
    c = synthetic_catalogue(20000, seed=0)            # (ships as data: ten minutes of numpy to draw)
    assert c.shape == (20000, 6) and np.isfinite(c).all()
    a, ecc, inc = elements(c)
    leo = int((a < RE_EQ + 2000e3).sum())
    geo = (np.abs(a - 42164e3) < 1.0) & (np.abs(inc) < 1e-8)
    equatorial, circular = int((np.abs(inc) < 1e-8).sum()), int((ecc < 1e-8).sum())
    molniya = int((np.abs(ecc - 0.737) < 1e-9).sum())
    tundra = int(((np.abs(ecc - 0.2) < 1e-9) & (np.abs(inc - np.radians(63.4)) < 1e-9)).sum())
    print("[catalogue] LEO %d (reference file 6 755)  equatorial %d (2 231)  circular %d (1 135)  Molniya %d  Tundra %d  ecc max %.3f" %
          (leo, equatorial, circular, molniya, tundra, ecc.max()))
    # regime shares = the draw probabilities 1/3, 1/3, 1/9, 1/9, 1/9 (orbit_gen.py:53): three binomial standard deviations
    for got, p in ((leo, 1 / 3), (equatorial, 1 / 9), (molniya, 1 / 9), (tundra, 1 / 9)):
        assert abs(got - 20000 * p) <= 3 * np.sqrt(20000 * p * (1 - p)), (got, p)
    assert abs(leo - 6755) <= 3 * 67 and abs(equatorial - 2231) <= 3 * 45          # the reference file's own counts
    assert equatorial == int(geo.sum())                                           # every GEO row is equatorial, nothing else is
    # circular rows: `stationary * ecc` (dynamics.py:383-385) -- the circular half of the GEO draws
    assert circular == int((geo & (ecc < 1e-8)).sum()) and abs(circular - 1135) <= 3 * 33 + 35
    assert ecc.max() <= 0.737 + 1e-9 and a.min() > RE_EQ + 300e3 and a.max() <= 42164e3 + 1.0


def test_every_row_passes_orbit_gens_acceptance_rule():
    """orbit_gen.py:55-70: above 300 km for four hours, and visible from the observer above 15 deg either all the time or within the first
    45 min with no gap of 1.5 h or more (150 s samples): every row of the shipped catalogue, and of a freshly drawn small one."""
    from datetime import datetime
    from ssa_gym_amd.envs.transformations import trans_matrix_table
    step, T = 150.0, 96
    M_t = trans_matrix_table(datetime(2020, 5, 4, 0, 0, 0), step, T)
    obs_lla = np.array((38.828198, -77.305352, 20.0)) * [host.deg2rad, host.deg2rad, 1]
    enu, obs_itrs = host.enu_matrix(obs_lla), host.lla2ecef(obs_lla)

    def rule(c):      # from the STATES (the catalogue holds no elements): propagate with the numpy Kepler of the generator
        a, ecc, inc = elements(c)
        r, v = c[:, :3], c[:, 3:]
        rn = np.linalg.norm(r, axis=1)
        evec = ((v * v).sum(1) - MU / rn)[:, None] * r / MU - (r * v).sum(1)[:, None] * v / MU
        h = np.cross(r, v)
        circ = ecc < 1e-8
        P = np.where(circ[:, None], r / rn[:, None], evec / np.maximum(ecc, 1e-300)[:, None])
        Q = np.cross(h / np.linalg.norm(h, axis=1)[:, None], P)
        nu = np.arctan2((r * Q).sum(1), (r * P).sum(1))
        E0 = 2.0 * np.arctan2(np.sqrt(1 - ecc) * np.sin(nu / 2), np.sqrt(1 + ecc) * np.cos(nu / 2))
        M0, n, b = E0 - ecc * np.sin(E0), np.sqrt(MU / a ** 3), a * np.sqrt(1 - ecc ** 2)
        ok_alt = np.ones(len(c), dtype=bool)
        vis = np.empty((T, len(c)), dtype=bool)
        for i in range(T):
            M = M0 + n * step * i
            E = M + ecc * np.sin(M)
            for _ in range(12):
                E = E - (E - ecc * np.sin(E) - M) / (1 - ecc * np.cos(E))
            x = ((a * (np.cos(E) - ecc))[:, None] * P + (b * np.sin(E))[:, None] * Q) @ M_t[i].T
            xn = np.linalg.norm(x, axis=1)
            ok_alt &= xn - cat_mod.WGS84_A * (1 - cat_mod.WGS84_F * (x[:, 2] / xn) ** 2) > 299e3
            d = x - obs_itrs
            vis[i] = np.arcsin((d @ enu[:, 2]) / np.linalg.norm(d, axis=1)) >= np.radians(15.0) - 1e-9
        run, worst = np.zeros(len(c), dtype=int), np.zeros(len(c), dtype=int)
        for i in range(T):
            run = np.where(vis[i], 0, run + 1)
            worst = np.maximum(worst, run)
        return ok_alt & (vis.all(0) | (vis[:18].any(0) & (worst < 36)))
    c = synthetic_catalogue(20000, seed=0)
    ok = rule(c)
    assert ok.mean() > 0.999, (~ok).sum()          # (a row exactly on the 15 deg / 300 km edge may flip with the rounding of the rebuilt elements)
    small = synthetic_catalogue(60, seed=3)          # drawn here, by the recipe (a few seconds)
    assert small.shape == (60, 6) and rule(small).all()
    free = synthetic_catalogue(2000, seed=3, visibility=False)
    assert rule(free).mean() < 0.5                   # (the rule is a real constraint: most unconstrained draws fail it)


def test_regime_order_is_a_permutation_sorted_by_semi_major_axis_and_dealt_over_the_xcds():
    """catalogue.regime_order: the layout hint of round 4 (objects of one regime share wavefronts; every XCD's run of tiles gets the same
    share of every regime).  A permutation; chunk c of the sorted list sits in tile (c % 8) * q + c // 8."""
    from ssa_gym_amd.catalogue import regime_order, MU
    cat = synthetic_catalogue(20000, seed=0)
    order = regime_order(cat)
    assert sorted(order.tolist()) == list(range(20000))
    a = 1.0 / (2.0 / np.linalg.norm(cat[:, :3], axis=1) - np.sum(cat[:, 3:] ** 2, axis=1) / MU)
    q = 5000 // 8
    chunks = np.empty(20000, dtype=np.int64)
    for c in range(5000):
        t = (c % 8) * q + c // 8
        chunks[4 * c:4 * c + 4] = order[4 * t:4 * t + 4]
    assert np.all(np.diff(a[chunks]) >= 0)                       # undone the dealing: ascending semi-major axis
    leo = a[order].reshape(5000, 4) < 8.4e6                      # LEO objects sit in whole tiles (all four or none, but for one boundary tile)
    assert np.sum(leo.any(axis=1) & ~leo.all(axis=1)) <= 1
    per_xcd = leo.all(axis=1).reshape(8, q).sum(axis=1)
    assert per_xcd.max() - per_xcd.min() <= 1
    odd = regime_order(cat[:1003])                               # not a multiple of 32: the plain sort
    assert sorted(odd.tolist()) == list(range(1003)) and np.all(np.diff(a[:1003][odd]) >= 0)


def test_regime_order_env_rotates_the_sorted_tiles():
    """catalogue.regime_order_env: env e of a vector env's batch gets the plain sort by semi-major axis rotated by e / n_env of its tiles
    (a wavefront of the batch's launch walks the same position of every env: rotated, every walk meets the same mix of regimes)"""
    from ssa_gym_amd.catalogue import regime_order_env, synthetic_catalogue, MU
    x = synthetic_catalogue(640, seed=3, visibility=False)
    a = 1.0 / (2.0 / np.linalg.norm(x[:, :3], axis=1) - np.sum(x[:, 3:] ** 2, axis=1) / MU)
    base = regime_order_env(x, 0, 8)
    assert np.array_equal(np.sort(base), np.arange(640)) and np.all(np.diff(a[base]) >= 0)
    for e in range(1, 8):
        o = regime_order_env(x, e, 8)
        assert np.array_equal(o, np.roll(base, -4 * ((e * 160) // 8)))
    # the slow regime (the lowest third of the semi-major axes) over the positions a wavefront walks: one tile per env at the same position
    slow = a < np.sort(a)[213]
    per_walk = sum(slow[regime_order_env(x, e, 8)].reshape(160, 4).any(axis=1).astype(int) for e in range(8))
    assert per_walk.min() >= 2 and per_walk.max() <= 4          # (sorted alike it would be 0 or 8)
    assert np.array_equal(regime_order_env(x[:10], 1, 8), np.argsort(a[:10], kind="stable"))      # (not whole tiles: the plain sort)
